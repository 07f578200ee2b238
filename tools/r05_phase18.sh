#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 23 24 25; do timeout 120 build/micro/conv_wino $i 9 | tail -1; done
for i in 7 13 10 17; do for v in 4 8 9; do timeout 120 build/micro/conv_wino $i $v | tail -2; done; done
} > gpurun_out/r05_micro_wino1d_kernel.txt 2>&1
cat gpurun_out/r05_micro_wino1d_kernel.txt
