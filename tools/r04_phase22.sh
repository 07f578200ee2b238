#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | cut -c1-110 | sed "s/^/$1: /"; }
for b in 64 96 128 144; do HCF_WG_BATCH_BLOCKS=$b run "batch $b     "; done
export HCF_WG_BATCH_BLOCKS=128
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python tools/train_bench.py --steps 2 --optim native > /dev/null 2> $O/prof_train.err
python tools/rocpd_trace.py /tmp/prof_train 3 2>/dev/null | grep "^#" > $O/trace_train_batch.txt
head -14 $O/trace_train_batch.txt | cut -c1-120; tail -3 $O/trace_train_batch.txt
