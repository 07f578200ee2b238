#!/bin/bash
# two-stream split as the module default: the whole -m gpu suite, then the driver's own bench invocation
O=gpurun_out/r05_p29
mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^shapes" > $O/pytest_full.log
grep -E "passed|failed|FAILED|rror" $O/pytest_full.log | tail -5
python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], "single", j.get("single_stream"), "frac", j["roofline"]["frac"], j["roofline"].get("avg_launch_us"))
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"].get("c_net"))
PY
