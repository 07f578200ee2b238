#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p4
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/train_bench.py --steps 6 > $O/train.txt 2>&1
tail -1 $O/train.txt
python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_callers.py tests/test_gpu_lu.py tests/test_gpu_gan.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -15 > $O/pytest.log
tail -4 $O/pytest.log
