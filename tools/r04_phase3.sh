#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p3
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wino.py tests/test_gpu_f16x3.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_wino.log
tail -3 $O/pytest_wino.log
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
for v in "" "HCF_NO_WINO_HALF=1"; do
  tag=${v:+nohalf}; tag=${tag:-half}
  env $v python bench.py --batch 1 --steps 40 --warmup 5 $COMMON > $O/b1_$tag.json 2>/dev/null
  env $v python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 30 --warmup 5 $COMMON > $O/c3_$tag.json 2>/dev/null
  env $v python tools/train_bench.py --steps 5 2>&1 | tail -1 > $O/train_$tag.txt
done
python - <<PY
import json
for t in ("half","nohalf"):
    for c in ("b1","c3"):
        j=json.loads(open("$O/%s_%s.json"%(c,t)).read().strip().splitlines()[-1])
        print(c,t,j["value"],j["ms_per_step"])
    print(open("$O/train_%s.txt"%t).read().strip())
PY
