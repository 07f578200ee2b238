#!/bin/bash
# N16 fused-tail conv with half the weight tile in LDS (4 blocks per CU): parity tests, then same-box A/B against the previous build
# (build/ab/libhcflow_hip_base.so through HCFLOW_LIB), one stream (per-kernel numbers) and the default
O=gpurun_out/r05_p36
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -5 > $O/pytest.log
grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -4
for rep in 1 2; do
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
