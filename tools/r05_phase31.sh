#!/bin/bash
# deferred parameter-stamp check: engine / nets / callers / real / full-size tests, then the bench line
O=gpurun_out/r05_p31
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_nets.py tests/test_gpu_callers.py tests/test_gpu_real.py tests/test_gpu_lu.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -6 > $O/pytest.log
grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -5
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], "single", j["single_stream"]["value"], "frac", j["roofline"]["frac"])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
PY
timeout 300 python tools/config3_split_probe.py 2>&1 | grep -E "^sync|^lazy" | tee $O/config3_split.txt
