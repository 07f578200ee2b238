#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p11
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_lu.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest.log
grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -12
for rep in 1 2; do
  python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | sed "s/^/strips  : /"
  HCF_NO_DG_STRIP=1 python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | sed "s/^/dgrad per-image: /"
done
