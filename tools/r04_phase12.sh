#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for wb in 256 192 160 144 128 96; do
  HCF_WG_BLOCKS=$wb python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | sed "s/^/wg_blocks $wb: /"
done
HCF_WG_BLOCKS=256 python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | sed "s/^/wg_blocks 256 again: /"
