#!/bin/bash
O=gpurun_out/r05_p46
mkdir -p $O
python tools/first_call_probe.py 2>&1 | grep -v "^shapes" | tail -2 | tee $O/first_call.txt
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_lu.py tests/test_gpu_nets.py tests/test_gpu_real.py tests/test_gpu_callers.py -m gpu -q -x 2>&1 | grep -v "^shapes" | grep -E "passed|failed|FAILED|rror" | tail -4 | tee $O/pytest.log
