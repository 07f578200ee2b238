#!/bin/bash
# re-entry state check (rebuilt .so): engine / full-size tests, default bench line and the two-stream line on the same box
O=gpurun_out/r05_p21
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -5 > $O/pytest.log
grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -3
for s in 1 2 1 2; do
  HCFLOW_STREAMS=$s python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check > $O/bench_s$s.json 2> $O/bench_s$s.err
  python - <<PY
import json
j=json.loads(open("$O/bench_s$s.json").read().strip().splitlines()[-1])
print("STREAMS $s BENCH", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("power"))
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
PY
done
