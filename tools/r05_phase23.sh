#!/bin/bash
# (a) which bench leg decides whether config 1 reads 13.6 or 22 ms afterwards; (b) the shared side-stream pool: config 5 after a
# two-stream config 2 in the same process; (c) the stream / training / engine tests with the pool
O=gpurun_out/r05_p23
mkdir -p $O
show() {
python - <<PY
import json
j=json.loads(open("$1").read().strip().splitlines()[-1])
print("$2", j["value"], j["ms_per_step"], {k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
PY
}
HCFLOW_STREAMS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precision --no-exact-check > $O/a.json 2> $O/a.err; show $O/a.json "S1 none"
HCFLOW_STREAMS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-check > $O/b.json 2> $O/b.err; show $O/b.json "S1 +other-precision"
HCFLOW_STREAMS=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precision > $O/c.json 2> $O/c.err; show $O/c.json "S1 +exact-check"
HCFLOW_STREAMS=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-precision --no-exact-check > $O/d.json 2> $O/d.err; show $O/d.json "S2 none"
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_backward.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -3
