#!/bin/bash
# the B = 1 slow mode inside the bench process: kernel time vs wall, clock, power (HCF_BENCH_B1_DIAG)
O=gpurun_out/r05_p25
mkdir -p $O
show() {
python - <<PY
import json
j=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", j["value"], j["other_configs"]["config1_single_patch_latency"])
PY
}
export HCF_BENCH_B1_DIAG=1
for st in 10 20; do
for s in 2 1; do
HCFLOW_STREAMS=$s python bench.py --steps $st --warmup 3 --no-cpu-baseline --no-other-precision --no-exact-check --other-steps 4 > $O/s${s}_$st.json 2> $O/s${s}_$st.err; show $O/s${s}_$st.json "S$s steps $st"
done
done
