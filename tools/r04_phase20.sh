#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for wb in 160 128 96 192 160; do
  HCF_WG_BLOCKS=$wb python tools/train_bench.py --steps 6 --optim native 2>&1 | tail -1 | cut -c1-110 | sed "s/^/th4 wg_blocks $wb: /"
done
