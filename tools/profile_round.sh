#!/bin/bash
# One GPU-box pass producing the per-round measurement artefacts (copied into profiles/ afterwards):
#   tools/profile_round.sh r02      -> gpurun_out/prof_r02/{bench.json, kernel_stats.txt, traffic_pmc.json, pmc_sq.json, *.log}
# rocprofv3: kernel trace and counters in SEPARATE runs (never --pmc together with sys/hip/hsa tracing), with HCFLOW_STREAMS=1: the
# module's default two-stream split overlaps kernels, and per-kernel durations / counters are only meaningful without it (bench.py's
# roofline block is taken from its single-stream region for the same reason).
set -u
R=${1:-rNN}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
HCFLOW_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $OUT/kt.log 2>&1
python $ROOT/tools/rocpd_summary.py $OUT/kt > $OUT/kernel_stats.txt 2>> $OUT/kt.log
rm -rf $OUT/kt
for c in FETCH_SIZE WRITE_SIZE; do
  HCFLOW_STREAMS=1 timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-check --no-other-precision --no-other-configs > $OUT/pmc_$c.log 2>&1
done
python $ROOT/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE > $OUT/traffic_pmc.json 2> $OUT/traffic.err
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/sq -- python $ROOT/tools/conv_bench.py --precision f16x3 --iters 3 > $OUT/sq.log 2>&1
python $ROOT/tools/pmc_sq.py $OUT/sq > $OUT/pmc_sq.json 2> $OUT/sq.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/sqx -- python $ROOT/tools/conv_bench.py --precision exact --iters 3 > $OUT/sqx.log 2>&1
python $ROOT/tools/pmc_sq.py $OUT/sqx > $OUT/pmc_sq_exact.json 2>> $OUT/sq.err
rm -rf $OUT/sq $OUT/sqx
ls -la $OUT
