#!/bin/bash
# kernel traces (rocpd) of B = 1, Face x8 and the training step -> gpurun_out/r04_small/{trace,kstats}_*.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check"
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -- "$@" > /dev/null 2> $O/prof_$name.err
  python tools/rocpd_trace.py /tmp/prof_$name $PARTS > $O/trace_$name.txt 2>> $O/prof_$name.err
  python tools/rocpd_summary.py /tmp/prof_$name > $O/kstats_$name.txt 2>> $O/prof_$name.err
}
PARTS=7 run b1 python bench.py --batch 1 --steps 5 --warmup 2 $COMMON
PARTS=7 run c3 python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 5 --warmup 2 $COMMON
PARTS=4 run train python tools/train_bench.py --steps 3
ls -la $O
