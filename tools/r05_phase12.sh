#!/bin/bash
# round 5, phase 12: the 16-wide channel tile of the fused-tail conv (HCF_NO_N16=1 = before): parity suites, then a same-box A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p12
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_f16x3.py tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -v "^shapes" | tail -12 > $O/pytest.log
tail -5 $O/pytest.log
COMMON="--no-other-precision --no-cpu-baseline --no-other-configs"
for rep in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export HCF_NO_N16=1; else unset HCF_NO_N16; fi
    python bench.py --steps 10 --warmup 3 $COMMON > $O/b_${v}_$rep.json 2>/dev/null
  done
done
unset HCF_NO_N16
python - <<PY
import json
for rep in (1,2):
    for v in (0,1):
        j=json.loads(open("$O/b_%d_%d.json"%(v,rep)).read().strip().splitlines()[-1])
        k=[x for x in j["roofline"]["conv_kernels"] if "TAILC" in x["kernel"]][0]
        print("no_n16",v,"rep",rep,j["value"],j["ms_per_step"],"tail",k["launches_per_step"],k["ms_per_step"],k["avg_launch_us"],k["frac_of_yardstick"],k["bound"],"f16x3 vs exact",j["precision"]["check"]["max_abs_diff_f16x3_vs_exact_f32"])
PY
