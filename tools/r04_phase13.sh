#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in 10 11 12 1 3; do
  HCF_NO_WINO=1 HCFLOW_LIB=hcflow_amd/libhcflow_hip_timers.so python tools/conv_bench.py --precision f16x3 --iters 20 --only $i 2>&1 | grep -v "^shapes\|amdgpu.ids"
done
for rep in 1 2; do python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1; done
