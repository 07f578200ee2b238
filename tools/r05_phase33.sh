#!/bin/bash
# the opt-out path (HCFLOW_STREAMS=1: one stream, one engine) stays green: the whole -m gpu suite under it
O=gpurun_out/r05_p33
mkdir -p $O
HCFLOW_STREAMS=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest_streams1.log
grep -E "passed|failed|FAILED|rror" $O/pytest_streams1.log | tail -5
