#!/bin/bash
O=gpurun_out/r05_p35
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -6 | tee $O/pytest.log
