#!/bin/bash
O=gpurun_out/r05_p44
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_engine.py -m gpu -q -x 2>&1 | grep -v "^shapes" | grep -E "passed|failed|FAILED|rror" | tail -4 | tee $O/pytest.log
