#!/bin/bash
# A/B: consecutive Winograd launches walk their units in opposite directions (HCF_WINO_REV=1, default) against all forward (0)
mkdir -p gpurun_out
run() { python bench.py --steps 8 --warmup 2 --no-other-precision --no-exact-check --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$1', d['value'], d['ms_per_step'], 'dominant', r['frac'], r['conv_kernels'][0]['ms_per_step'], [ (k['kernel'][:30], k['ms_per_step']) for k in r['conv_kernels'][1:4]])"; }
{
for rep in 1 2; do
  HCF_WINO_REV=0 run rev0
  HCF_WINO_REV=1 run rev1
done
} > gpurun_out/r05_ab_wino_rev.txt 2>&1
cat gpurun_out/r05_ab_wino_rev.txt
