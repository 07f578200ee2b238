#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p6
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  python tools/train_bench.py --steps 6 2>&1 | tail -1 | sed 's/^/side-stream: /'
  HCF_NO_WGRAD_STREAM=1 python tools/train_bench.py --steps 6 2>&1 | tail -1 | sed 's/^/one stream : /'
done
for p in 192 384; do HCF_EPI_PPB=$p python tools/train_bench.py --steps 6 2>&1 | tail -1 | sed "s/^/ppb $p: /"; done
python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_callers.py tests/test_gpu_lu.py -m gpu -q 2>&1 | tail -6 > $O/pytest.log
grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
