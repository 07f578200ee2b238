#!/bin/bash
# round 5, phase 5: the top-of-unit vmcnt(0) (which sits out the previous epilogue's store acknowledgements) moved in front of the
# epilogue's first store: baseline binary against the new one, isolated loops
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p5
mkdir -p $O
cd $GRAFT_REPO_ROOT
{
echo "== checks (new)"
for i in 0 1 2 3 18 19; do timeout 120 build/micro/conv_wino_nowait $i 4; done
for rep in 1 2; do
  for b in conv_wino conv_wino_nowait; do
    echo "== $b"
    for i in 4 5 6 7 9 10 11 12 13 15 16 17 20 21; do timeout 120 build/micro/$b $i 4; done
  done
done
} > $O/micro_nowait.txt 2>&1
cat $O/micro_nowait.txt
