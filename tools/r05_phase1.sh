#!/bin/bash
# round 5, phase 1: v6 (row phase pipelined under the position loop) against v4, isolated; this round's baseline bench; new parity tests
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p1
mkdir -p $O
cd $GRAFT_REPO_ROOT
M=build/micro/conv_wino
{
for v in 4 6 7; do
  echo "== version $v checks"
  for i in 1 18 19; do timeout 120 $M $i $v; done
done
for v in 4 6 7 4 6 7; do
  echo "== version $v timings"
  for i in 7 10 11 13 15 17 20 21; do timeout 120 $M $i $v; done
done
} > $O/micro_v6.txt 2>&1
{
for v in 4 6 7; do
  echo "== version $v prof"
  for i in 7 13 17; do timeout 120 build/micro/conv_wino_prof $i $v; done
done
} > $O/micro_v6_prof.txt 2>&1
cat $O/micro_v6.txt
cat $O/micro_v6_prof.txt
python bench.py --steps 10 --warmup 3 --no-other-configs > $O/bench_base.json 2> $O/bench_base.err
tail -c 600 $O/bench_base.json | head -c 300; echo
python - <<PY
import json
j=json.loads(open("$O/bench_base.json").read().strip().splitlines()[-1])
print("BASE", j["value"], j["ms_per_step"])
for k in j["roofline"]["conv_kernels"]: print(k["kernel"][:60], k["launches_per_step"], k["ms_per_step"], k["avg_launch_us"], k["tflops"])
PY
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_lu.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_new.log
cat $O/pytest_new.log
