#!/bin/bash
O=gpurun_out/r05_p42
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -x 2>&1 | grep -v "^shapes" | grep -E "passed|failed|FAILED|rror|assert" | tail -8 | tee $O/pytest.log
