#!/bin/bash
# final evidence of round 5 (second session): the whole -m gpu suite, the profiling pass (bench line, rocprofv3 kernel stats, PMC
# traffic, SQ counters; profiling legs under HCFLOW_STREAMS=1), the config-5 line, the kernel stats of the training step
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final2
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED|error" $O/pytest.log | tail -5
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 > $O/train_bench_prof.txt 2> $O/prof_train.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_train > $O/kstats_train.txt 2>> $O/prof_train.err
cd $GRAFT_REPO_ROOT
python - <<PY
import json
j=json.loads(open("gpurun_out/prof_r05/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], j["single_stream"]["value"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"]["avg_launch_us"])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
t=json.loads(open("gpurun_out/r05_final2/train_line.json").read().strip().splitlines()[-1])
print("TRAIN", t["value"], t["ms_per_step"], t.get("other_optimizer",{}).get("ms_per_step"))
PY
