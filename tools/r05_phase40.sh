#!/bin/bash
# config 5: the reference's loop on the process' default stream against the same loop inside a side stream (experiment knob)
O=gpurun_out/r05_p40
mkdir -p $O
for rep in 1 2; do
for side in 0 1; do
  if [ $side = 1 ]; then export HCF_BENCH_SIDE_STREAM=1; else unset HCF_BENCH_SIDE_STREAM; fi
  python bench.py --workload train --steps 12 --warmup 3 > $O/train_$side.json 2> $O/train_$side.err
  python - <<PY
import json
t=json.loads(open("$O/train_$side.json").read().strip().splitlines()[-1])
print("side stream $side:", t["value"], t["ms_per_step"], "native", t.get("other_optimizer",{}).get("ms_per_step"))
PY
done
done
