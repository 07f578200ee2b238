#!/bin/bash
# weight-gradient kernels, interleaved form with FOUR threads per pixel (half the staging slots): parity + bit-identity, kernel stats
# against the single-buffer form on the same box, the config-5 line
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p48
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py tests/test_gpu_lu.py tests/test_gpu_callers.py tests/test_gpu_fuzz.py tests/test_gpu_gan.py tests/test_gpu_optim.py tests/test_gpu_engine.py -x -q 2>&1 | grep -v "^shapes" | tail -12 > $O/pytest.log
grep -E "passed|failed|FAILED|rror" $O/pytest.log | tail -4
cd /tmp
for mode in single inter; do
  unset HCF_WG_SINGLE_BUF
  [ "$mode" = single ] && export HCF_WG_SINGLE_BUF=1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 4 > $O/prof_$mode.txt 2> $O/prof_$mode.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_$mode > $O/kstats_$mode.txt 2>> $O/prof_$mode.err
  echo "== $mode: $(tail -1 $O/prof_$mode.txt)"; grep -E "wgrad|total kernel" $O/kstats_$mode.txt | cut -c1-150
done
unset HCF_WG_SINGLE_BUF
cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
python - <<PY
import json
t=json.loads(open("gpurun_out/r05_p48/train_line.json").read().strip().splitlines()[-1])
print("TRAIN", t["value"], t["ms_per_step"], t.get("other_optimizer",{}).get("ms_per_step"))
PY
