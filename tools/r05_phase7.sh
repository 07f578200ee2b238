#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p7
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -m gpu -q -x -s -k "forced or overflow" 2>&1 | grep -v "^shapes\|^\.shapes" | tail -30
