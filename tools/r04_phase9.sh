#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p9
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_optim.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest.log
grep -E "passed|failed|FAILED|Error|assert" $O/pytest.log | tail -12
for o in torch native torch-fused; do
  python tools/train_bench.py --steps 6 --optim $o 2>&1 | tail -1 | sed "s/^/$o: /"
done
