#!/bin/bash
# weight-gradient kernels: the double-buffered LDS forms against the single-buffer form (parity, bit-identity, same-box A/B of the
# training step's backward phase, kernel stats of the single-buffer and the interleaved form)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p47
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 420 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_full.py -x -q 2>&1 | grep -v "^shapes" | tail -12 > $O/pytest.log
tail -4 $O/pytest.log
run() {  # mode
  unset HCF_WG_SINGLE_BUF HCF_WG_DB_BLOCK
  [ "$1" = single ] && export HCF_WG_SINGLE_BUF=1
  [ "$1" = block ] && export HCF_WG_DB_BLOCK=1
}
for mode in single inter block single inter; do
  run $mode
  echo "== $mode: $(timeout 200 python tools/train_bench.py --steps 8 2>/dev/null | tail -1)" | tee -a $O/ab.txt
done
cd /tmp
for mode in single inter; do
  run $mode
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -- python $GRAFT_REPO_ROOT/tools/train_bench.py --steps 4 > $O/prof_$mode.txt 2> $O/prof_$mode.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/prof_$mode > $O/kstats_$mode.txt 2>> $O/prof_$mode.err
  echo "== $mode"; grep -E "wgrad|total kernel" $O/kstats_$mode.txt | cut -c1-150
done
