#!/bin/bash
# final evidence of the round: bench line, kernel stats of the bench and of the training step, the whole -m gpu suite
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_final
mkdir -p $O
cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python tools/train_bench.py --steps 6 > $O/train.txt 2>&1
python tools/train_bench.py --steps 6 --optim native > $O/train_native.txt 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -- python tools/train_bench.py --steps 3 --optim native > /dev/null 2> $O/prof_train.err
python tools/rocpd_summary.py /tmp/prof_train > $O/kstats_train.txt 2>> $O/prof_train.err
python tools/rocpd_trace.py /tmp/prof_train 4 2>/dev/null | grep "^#" > $O/trace_train_summary.txt
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest.log
grep -E "passed|failed|FAILED" $O/pytest.log | tail -5
tail -1 $O/train.txt
tail -1 $O/train_native.txt
