#!/bin/bash
# final evidence of round 5 (fourth session, after the weight-gradient LDS forms): the whole -m gpu suite, the bench line, per-launch
# traces of B = 1 and Face x8 at the end of the round (single stream: HCFLOW_STREAMS=1), the config-5 line and the training step's
# kernel stats. (Inference kernels are unchanged since tools/r05_final2.sh: its PMC / SQ files stay the round's.)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final3
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^shapes" | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED|error" $O/pytest.log | tail -5
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
timeout 200 python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs --no-single-stream-leg"
run() { name=$1; shift
  HCFLOW_STREAMS=1 timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- "$@" > /dev/null 2> $O/prof_$name.err
  python tools/rocpd_trace.py /tmp/prof_$name $PARTS > $O/trace_$name.txt 2>> $O/prof_$name.err
  python tools/rocpd_summary.py /tmp/prof_$name > $O/kstats_$name.txt 2>> $O/prof_$name.err
}
PARTS=7 run b1 python bench.py --batch 1 --steps 5 --warmup 2 $COMMON
PARTS=7 run c3 python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 5 --warmup 2 $COMMON
cd /tmp
timeout 240 rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 5 > $O/train_bench_prof.txt 2> $O/prof_train.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_train > $O/kstats_train.txt 2>> $O/prof_train.err
cd $GRAFT_REPO_ROOT
python - <<PY
import json
j=json.loads(open("gpurun_out/r05_final3/bench.json").read().strip().splitlines()[-1])
print("BENCH", j["value"], j["ms_per_step"], j["single_stream"]["value"], j["roofline"]["frac"], j["roofline"]["bound"], j["roofline"]["avg_launch_us"])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in j.get("other_configs",{}).items() if isinstance(v,dict)})
t=json.loads(open("gpurun_out/r05_final3/train_line.json").read().strip().splitlines()[-1])
print("TRAIN", t["value"], t["ms_per_step"], t.get("other_optimizer",{}).get("ms_per_step"))
PY
