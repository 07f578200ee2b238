#!/bin/bash
# bench line with other_configs, the redone grid-barrier probe, the whole -m gpu suite (no -x)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_p1
mkdir -p $O
cd $GRAFT_REPO_ROOT
build/micro/grid_sync > $O/grid_sync.txt 2>&1
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.log
tail -3 $O/pytest.log
