#!/bin/bash
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -- python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 5 --warmup 2 $COMMON > /dev/null 2> $O/prof_c3.err
python tools/rocpd_summary.py /tmp/prof_c3 > $O/kstats_c3_end.txt 2>> $O/prof_c3.err
grep -E "step_tail|fcn12|TAIL|gauss" $O/kstats_c3_end.txt | cut -c1-140
python tools/rocpd_trace.py /tmp/prof_c3 7 2>/dev/null > $O/trace_c3_end.txt
