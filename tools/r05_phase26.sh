#!/bin/bash
# after the flat stamp + GC-quiet timed regions: the driver's own invocation, twice (is config 1 stable now?), engine tests
O=gpurun_out/r05_p26
mkdir -p $O
show() {
python - <<PY
import json
j=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", j["value"], j["ms_per_step"], j["roofline"]["frac"], {k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
c=j.get("cpu_baseline") or {}
print("   cpu", c.get("value"), c.get("threads"), c.get("c_net"))
PY
}
python bench.py > $O/a.json 2> $O/a.err; show $O/a.json "default"
python bench.py --steps 20 --warmup 4 --no-cpu-baseline > $O/b.json 2> $O/b.err; show $O/b.json "steps20"
HCFLOW_STREAMS=2 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check > $O/c.json 2> $O/c.err; show $O/c.json "S2 steps20"
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_nets.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -3
