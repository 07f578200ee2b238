#!/bin/bash
# 1-D Winograd kernel (v8) against v4: checks, then timings of the 64-output-channel shapes
mkdir -p gpurun_out
{
for i in 1 23 24 25; do timeout 120 build/micro/conv_wino $i 8; done
for i in 7 11 13 10 15 17 22; do timeout 120 build/micro/conv_wino $i 4; timeout 120 build/micro/conv_wino $i 8; done
echo "--- WINO_PROF builds"
for i in 7 11 13 10; do timeout 120 build/micro/conv_wino_prof $i 4; timeout 120 build/micro/conv_wino_prof $i 8; done
} > gpurun_out/r05_micro_wino1d_kernel.txt 2>&1
cat gpurun_out/r05_micro_wino1d_kernel.txt
