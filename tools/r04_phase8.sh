#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p8
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_callers.py tests/test_gpu_lu.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest.log
grep -E "passed|failed|FAILED|Error" $O/pytest.log | tail -8
for rep in 1 2; do
  python tools/train_bench.py --steps 6 2>&1 | tail -1 | sed 's/^/fused   : /'
  HCF_NO_EPI_FUSE=1 python tools/train_bench.py --steps 6 2>&1 | tail -1 | sed 's/^/separate: /'
done
