#!/bin/bash
# package power and average shader clock (rocm-smi) while bench.py's sampling pass runs for ~20 s
mkdir -p gpurun_out
( python bench.py --steps 160 --warmup 3 --no-other-precision --no-exact-check --no-cpu-baseline --no-other-configs > gpurun_out/r05_power_bench.json 2>/dev/null ) &
BP=$!
for k in $(seq 1 70); do
  echo -n "t=$((k))x0.5s "; rocm-smi --showpower --showclocks 2>&1 | grep -i "Package Power\|sclk" | sed 's/^.*: //' | tr '\n' ' '; echo
  kill -0 $BP 2>/dev/null || break
  sleep 0.35
done
wait $BP
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_power_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
