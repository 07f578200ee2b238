#!/bin/bash
# round 5, phase 15: the -m gpu suite of the inference modules with the opt-in two-stream split as the process default; then the whole suite with defaults
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p15
mkdir -p $O
cd $GRAFT_REPO_ROOT
HCFLOW_STREAMS=2 timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_nets.py tests/test_gpu_fullsize.py tests/test_gpu_real.py tests/test_gpu_callers.py tests/test_gpu_f16x3.py -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest_streams2.log
tail -4 $O/pytest_streams2.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest_all.log
tail -3 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
