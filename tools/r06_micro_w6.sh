#!/bin/bash
# round 6: the F(4x4,3x3) kernel (version 10 of tools/micro/conv_wino.hip, tools/micro/hcf_conv_wino6.h) against v4:
# correctness checks, the config-2 shapes, the per-phase profile (-DW6_PROF build) and the timing ablations (-DW6_ABL=n builds).
#   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I hcflow_amd/csrc -I tools/micro -I include [-DW6_PROF | -DW6_ABL=n] tools/micro/conv_wino.hip -o build/micro/conv_wino[_prof | _abl<n>]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r06_micro_w6}.txt
: > $OUT
for p in 1 23 24 25 35 36 37 38 39; do timeout 120 build/micro/conv_wino $p 10 >> $OUT 2>&1; done
for p in 7 13 11 10 17 15 26 27; do
  for v in 4 10; do echo -n "v$v " >> $OUT; timeout 120 build/micro/conv_wino $p $v >> $OUT 2>&1; done
done
echo "---- per-phase profile (W6_PROF build)" >> $OUT
for p in 7 13 11 10; do timeout 120 build/micro/conv_wino_prof $p 10 >> $OUT 2>&1; done
echo "---- timing ablations (results wrong by construction): 1 no weight loads, 2 no image DMA, 4 no MFMAs, 8 no patch reads, 16 no outputs, 32 no LDS accumulators, 63 all" >> $OUT
for p in 7 11; do for b in build/micro/conv_wino_abl*; do echo -n "abl ${b##*abl}: " >> $OUT; timeout 120 $b $p 10 >> $OUT 2>&1; done; done
cat $OUT
