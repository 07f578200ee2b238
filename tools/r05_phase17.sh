#!/bin/bash
# v8 timing ablations (results wrong by construction): 1 epilogue, 2 image DMA, 4 weight DMA, 8 barrier, 16 transform,
# 32 image always of units 0..7 (L2-resident), 64 image addressed as a channel-blocked layout (whole lines)
mkdir -p gpurun_out
{
for i in 7 13; do
  echo "== full"; timeout 120 build/micro/conv_wino $i 8
  for abl in 2 32 64 96; do echo "== W8_ABL=$abl"; timeout 120 build/micro/conv_wino_abl$abl $i 8; done
done
} > gpurun_out/r05_micro_wino1d_ablate2.txt 2>&1
cat gpurun_out/r05_micro_wino1d_ablate2.txt
