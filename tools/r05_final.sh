#!/bin/bash
# final evidence of round 5: the profiling pass (bench line, rocprofv3 kernel stats, PMC traffic, SQ counters), the config-5 line,
# the kernel stats of the training step, the whole -m gpu suite
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final
mkdir -p $O
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 > $O/train_bench_prof.txt 2> $O/prof_train.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_train > $O/kstats_train.txt 2>> $O/prof_train.err
python $GRAFT_REPO_ROOT/tools/rocpd_trace.py /tmp/prof_train 6 2>/dev/null | grep "^#" > $O/trace_train_last_step.txt
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED|error" $O/pytest.log | tail -5
python - <<PY
import json
j=json.loads(open("gpurun_out/prof_r05/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"]["avg_launch_us"])
PY
