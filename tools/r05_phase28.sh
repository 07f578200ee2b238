#!/bin/bash
O=gpurun_out/r05_p28
mkdir -p $O
timeout 600 python tools/config3_split_probe.py 2>&1 | grep -v "^shapes" | tail -10 | tee $O/config3_split.txt
