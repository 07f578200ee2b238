#!/bin/bash
# round 5, phase 3: config-5 harness (bench.py --workload train) at N = 1, free-running steps; one steady-state step's launch census
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p3
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --workload train --steps 10 --warmup 3 > $O/train_line.json 2> $O/train_line.err
tail -c 2500 $O/train_line.json; echo
tail -5 $O/train_line.err
cd /tmp
rocprofv3 --kernel-trace -d /tmp/prof_train -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 7 > $O/train_bench.txt 2> $O/prof_train.err
tail -2 $O/train_bench.txt
python $GRAFT_REPO_ROOT/tools/rocpd_trace.py /tmp/prof_train 8 2>/dev/null | grep "^#" > $O/trace_train_last_step.txt
head -40 $O/trace_train_last_step.txt
