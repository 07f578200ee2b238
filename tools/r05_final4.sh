#!/bin/bash
# HEAD of the fourth session: the whole -m gpu suite once more, the driver's default bench command, and the 1x1 weight-gradient
# kernels under the interleaved form (HCF_WG_INTERLEAVE_ALL=1) against their single-buffer default
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final4
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED|error" $O/pytest.log | tail -5
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
j=json.loads(open("gpurun_out/r05_final4/bench_default.json").read().strip().splitlines()[-1])
print("BENCH default flags", j["value"], j["ms_per_step"], j["steps"], j["warmup"], j["single_stream"]["value"], j["roofline"]["frac"])
PY
cd /tmp
for mode in default all; do
  unset HCF_WG_INTERLEAVE_ALL
  [ "$mode" = all ] && export HCF_WG_INTERLEAVE_ALL=1
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 4 > $O/prof_$mode.txt 2> $O/prof_$mode.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_$mode > $O/kstats_$mode.txt 2>> $O/prof_$mode.err
  echo "== $mode: $(tail -1 $O/prof_$mode.txt)"; grep -E "wgrad_f16x3_kernel<1" $O/kstats_$mode.txt | cut -c1-150
done
