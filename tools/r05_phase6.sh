#!/bin/bash
# round 5, phase 6: same-box A/B of the whole bench: top-of-unit wait of the 64-channel Winograd kernel (HCF_WINO_TOP_WAIT=1 = before)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p6
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
for rep in 1 2; do
  for v in 0 1; do
    HCF_WINO_TOP_WAIT=$v python bench.py --steps 10 --warmup 3 $COMMON > $O/b_${v}_$rep.json 2>/dev/null
  done
done
python - <<PY
import json
for rep in (1,2):
    for v in (0,1):
        j=json.loads(open("$O/b_%d_%d.json"%(v,rep)).read().strip().splitlines()[-1])
        k=[x for x in j["roofline"]["conv_kernels"] if "wino4_kernel<0|1|2>" in x["kernel"]][0]
        print("top_wait",v,"rep",rep,j["value"],j["ms_per_step"],"wino4",k["ms_per_step"],k["avg_launch_us"],k["frac_of_yardstick"],k["bound"])
PY
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k "forced" 2>&1 | grep -v "^shapes\|^\.shapes" | tail -4
