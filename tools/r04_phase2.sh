#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p2
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/train_bench.py --steps 6 > $O/train_gather.txt 2>&1
HCF_NO_DGRAD_GATHER=1 python tools/train_bench.py --steps 6 > $O/train_scatter.txt 2>&1
tail -2 $O/train_gather.txt $O/train_scatter.txt
python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_callers.py tests/test_gpu_lu.py tests/test_gpu_f16x3.py tests/test_gpu_ops.py tests/test_gpu_wino.py tests/test_gpu_nets.py tests/test_gpu_gan.py -m gpu -q 2>&1 | tail -40 > $O/pytest.log
tail -4 $O/pytest.log
