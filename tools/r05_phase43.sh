#!/bin/bash
# per-thread reverse-walk flip (the two enqueuing threads of a split call no longer share one counter): same-box A/B against the previous build
# (build/ab/libhcflow_hip_base.so through HCFLOW_LIB), one stream (per-kernel numbers) and the default
O=gpurun_out/r05_p43
mkdir -p $O
for rep in 1 2 3; do
for lib in base new; do
  if [ $lib = base ]; then export HCFLOW_LIB=$GRAFT_REPO_ROOT/build/ab/libhcflow_hip_base.so; else unset HCFLOW_LIB; fi
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check --no-other-configs > $O/$lib.json 2> $O/$lib.err
  python - <<PY
import json
j=json.loads(open("$O/$lib.json").read().strip().splitlines()[-1])
fam=[v for v in j["roofline"]["conv_kernels"] if "TAILC" in v["kernel"]]
print("$lib:", j["value"], j["ms_per_step"], "single", j["single_stream"]["value"], "tail family", [(v["ms_per_step"], v["avg_launch_us"]) for v in fam])
PY
done
done
