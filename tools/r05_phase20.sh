#!/bin/bash
# state check after the reverse walk + power sampler: full -m gpu suite, default bench line, config-5 line
O=gpurun_out/r05_p20
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -5
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"].get("power"))
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
t=json.loads(open("$O/train_line.json").read().strip().splitlines()[-1])
print("TRAIN", t["value"], t["ms_per_step"], t["roofline"].get("power"), t["other_optimizer"].get("ms_per_step"))
PY
