#!/bin/bash
O=gpurun_out/r05_p37
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | grep -E "passed|failed|FAILED|rror" | tail -6 | tee $O/pytest.log
