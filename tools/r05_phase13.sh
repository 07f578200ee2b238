#!/bin/bash
# round 5, phase 13: weight fragments of the 64-channel Winograd kernel read one position earlier (-DWINO4_PREFETCH), isolated A/B
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p13
mkdir -p $O
cd $GRAFT_REPO_ROOT
{
echo "== checks (prefetch)"
for i in 1 18 19; do timeout 120 build/micro/conv_wino_pf $i 4; done
for rep in 1 2 3; do
  for b in conv_wino conv_wino_pf; do
    echo "== $b"
    for i in 7 10 11 13 15 17 20 21; do timeout 120 build/micro/$b $i 4; done
  done
done
} > $O/micro_prefetch.txt 2>&1
cat $O/micro_prefetch.txt
