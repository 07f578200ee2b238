#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python tools/train_bench.py --steps 2 --optim native > /dev/null 2> $O/prof_train.err
python tools/rocpd_trace.py /tmp/prof_train 3 > $O/trace_train2.txt 2>> $O/prof_train.err
python tools/rocpd_summary.py /tmp/prof_train > $O/kstats_train2.txt 2>> $O/prof_train.err
grep "^#" $O/trace_train2.txt | head -30
