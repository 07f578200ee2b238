#!/bin/bash
# same-box A/B of an environment knob on the headline bench: tools/r04_ab.sh VAR=1 [steps]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
K=$1; S=${2:-10}
O=$GRAFT_REPO_ROOT/gpurun_out/r04_ab
mkdir -p $O
COMMON="--no-other-precision --no-cpu-baseline --no-other-configs"
for rep in 1 2; do
  python bench.py --steps $S --warmup 3 $COMMON > $O/base_$rep.json 2>/dev/null
  env $K python bench.py --steps $S --warmup 3 $COMMON > $O/knob_$rep.json 2>/dev/null
done
python - <<PY
import json
for t in ("base_1","knob_1","base_2","knob_2"):
    j=json.loads(open("$O/%s.json"%t).read().strip().splitlines()[-1])
    ck=j["precision"]["check"]
    print(t, j["value"], j["ms_per_step"], "diff_vs_exact", ck.get("max_abs_diff_f16x3_vs_exact_f32"), [ (k["kernel"][:40],k["launches_per_step"],k["ms_per_step"]) for k in j["roofline"]["conv_kernels"][:6]])
PY
