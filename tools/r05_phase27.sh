#!/bin/bash
# per-kernel view of config 4 (rescaling round trip, B = 8 at 640^2) and config 3 (Face x8, B = 32)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p27
mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --preset Rescaling_DF2K_4X --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-other-precision --no-exact-check --no-two-streams > $O/c4.json 2> $O/c4.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_c4 > $O/kernel_stats_config4.txt 2>> $O/c4.err
head -40 $O/kernel_stats_config4.txt | cut -c1-175
