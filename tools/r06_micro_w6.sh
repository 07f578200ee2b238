#!/bin/bash
# round 6: F(4x4,3x3) kernel (version 10 of tools/micro/conv_wino.hip) against v4 -- correctness checks, then the config-2 shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r06_micro_w6}.txt
: > $OUT
for p in 1 23 24 25; do timeout 120 build/micro/conv_wino $p 10 >> $OUT 2>&1; done
for p in 7 13 11 10 17 15 26 27; do
  for v in 4 10; do echo "-- version $v" >> $OUT; timeout 120 build/micro/conv_wino $p $v >> $OUT 2>&1; done
done
cat $OUT
