#!/bin/bash
# for the record: rocprofv3 kernel stats of the DEFAULT command (two streams: kernels of the two half batches overlap, durations inflate)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p34
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_def -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --no-other-precision --no-exact-check --no-single-stream-leg > $O/bench_default.json 2> $O/err.txt
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_def > $O/kernel_stats_default_two_streams.txt 2>> $O/err.txt
head -12 $O/kernel_stats_default_two_streams.txt | cut -c1-150
python - <<PY
import json
j=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("BENCH under rocprof", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"])
PY
