#!/bin/bash
# round 5, phase 4: per-sample range fallback tests; config-5 line with one module for both optimiser variants
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p4
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -m gpu -q -x -s 2>&1 | grep -v "^shapes" | tail -25 > $O/pytest.log
cat $O/pytest.log
python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
python - <<PY
import json
j=json.loads(open("$O/train_line.json").read().strip().splitlines()[-1])
print("train torch", j["value"], j["ms_per_step"], j["detail"]["phases_ms_one_synchronised_step"])
print("train other", j["other_optimizer"].get("value"), j["other_optimizer"].get("ms_per_step"), j["other_optimizer"].get("phases_ms_one_synchronised_step"), j["other_optimizer"].get("error"))
PY
python bench.py --workload train --optim native --steps 10 --warmup 3 > $O/train_line_native_first.json 2>> $O/train_line.err
python - <<PY
import json
j=json.loads(open("$O/train_line_native_first.json").read().strip().splitlines()[-1])
print("train native-first", j["value"], j["ms_per_step"], j["detail"]["phases_ms_one_synchronised_step"])
print("train other", j["other_optimizer"].get("value"), j["other_optimizer"].get("ms_per_step"), j["other_optimizer"].get("phases_ms_one_synchronised_step"), j["other_optimizer"].get("error"))
PY
