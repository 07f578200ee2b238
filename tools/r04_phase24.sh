#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p24
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_lu.py tests/test_gpu_callers.py tests/test_gpu_gan.py -m gpu -q -x > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -8
run() { python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | cut -c1-110 | sed "s/^/$1: /"; }
for rep in 1 2; do
run "third stream  "
HCF_NO_DG_STREAM=1 run "chain only    "
done
