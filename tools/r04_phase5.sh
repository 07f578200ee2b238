#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p5
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 30 --warmup 5 $COMMON > $O/c3.json 2>/dev/null
python tools/train_bench.py --steps 6 2>&1 | tail -1 > $O/train.txt
python - <<PY
import json
j=json.loads(open("$O/c3.json").read().strip().splitlines()[-1]); print("c3", j["value"], j["ms_per_step"])
print(open("$O/train.txt").read().strip())
PY
python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest.log
grep -E "passed|failed|FAILED" $O/pytest.log | tail -8
