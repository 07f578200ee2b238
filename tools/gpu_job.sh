#!/bin/bash
# tools/gpu_job.sh <name> <command...>: run one command from the repo root on the GPU box, log under gpurun_out/<name>.log
#   gpurun --timeout 900 -- 'bash tools/gpu_job.sh ab_two_streams env HCFLOW_STREAMS=1 python bench.py --steps 10 --no-cpu-baseline'
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
name=$1; shift
"$@" > gpurun_out/$name.log 2>&1
rc=$?
tail -n 40 gpurun_out/$name.log
exit $rc
