#!/bin/bash
# B = 1 bimodality, second pass (conditions of the bench process: the big net kept alive, more steps), then the small-piece probe
O=gpurun_out/r05_p24
mkdir -p $O
for c in after_b16_keep after_b16_s2_keep; do
  for st in 13 24; do
    B16_STEPS=$st timeout 300 python tools/b1_probe.py $c 2>&1 | grep -v "^shapes" | tail -5 | sed "s/^/steps $st  /" | tee -a $O/b1_probe.txt
  done
done
timeout 600 python tools/multi_stream_probe.py 2>&1 | grep -v "^shapes" | tail -12 | tee $O/multi_stream.txt
