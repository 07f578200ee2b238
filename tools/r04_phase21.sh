#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p21
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_lu.py -m gpu -q -x > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -8
run() { python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | cut -c1-110 | sed "s/^/$1: /"; }
HCF_NO_WG_BATCH=1 run "per conv      "
for b in 128 160 208 256 320; do HCF_WG_BATCH_BLOCKS=$b run "batch $b     "; done
HCF_NO_WG_BATCH=1 run "per conv again"
