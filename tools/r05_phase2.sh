#!/bin/bash
# round 5, phase 2: is the image DMA's cost its access pattern? NHWC (64 B of a 256 / 768-byte pixel record per lane group) against a
# 16-channel-blocked layout (contiguous 64-byte pixels), and the nt cache policy on the image pieces. Timing only (v4 kernel).
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p2
mkdir -p $O
cd $GRAFT_REPO_ROOT
{
timeout 120 build/micro/conv_wino 1 4
for rep in 1 2; do
  for mode in nhwc blocked nt; do
    echo "== $mode"
    for i in 7 10 11 13 15 17; do
      if [ $mode = blocked ]; then timeout 120 build/micro/conv_wino_blk $i 4
      elif [ $mode = nt ]; then timeout 120 build/micro/conv_wino_nt $i 4
      else timeout 120 build/micro/conv_wino $i 4; fi
    done
  done
done
} > $O/micro_blocked.txt 2>&1
cat $O/micro_blocked.txt
