#!/bin/bash
# the B = 1 bimodality (13.6 vs 22-26 ms per call): one condition per process
O=gpurun_out/r05_p22
mkdir -p $O
for c in fresh lazy side_stream after_b16 after_b16_s2 after_b16_side fresh; do
  timeout 300 python tools/b1_probe.py $c 2>&1 | grep -v "^shapes" | tail -4 | tee -a $O/b1_probe.txt
done
