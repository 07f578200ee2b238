#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python bench.py --preset Rescaling_DF2K_4X --batch 8 --lr-size 160 --steps 5 --warmup 2 $COMMON > /dev/null 2> $O/prof_c4.err
python tools/rocpd_summary.py /tmp/prof_c4 > $O/kstats_c4.txt 2>> $O/prof_c4.err
python tools/rocpd_trace.py /tmp/prof_c4 7 > $O/trace_c4.txt 2>> $O/prof_c4.err
head -30 $O/kstats_c4.txt | cut -c1-150
