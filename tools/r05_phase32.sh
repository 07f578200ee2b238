#!/bin/bash
# soak of the new default (two streams + helper thread + deferred check): determinism at the bench workload, both precisions, training gradients
O=gpurun_out/r05_p32
mkdir -p $O
timeout 900 python tools/stress_determinism.py --passes 60 2>&1 | grep -v "^shapes" | tail -6 | tee $O/stress_determinism.txt
