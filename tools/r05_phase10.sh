#!/bin/bash
# why do the two half-batch streams not overlap inside bench.py? per-stream timeline of two steps; B = 1 under both settings; both halves off the null stream
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_p10
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check --no-other-configs"
for v in 2 1; do
  HCFLOW_STREAMS=$v python bench.py --batch 1 --steps 20 --warmup 5 $COMMON 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=1 streams $v', j['ms_per_step'])"
done
HCFLOW_SPLIT_BOTH_SIDE=1 python bench.py --steps 8 --warmup 3 $COMMON 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('both-side', j['value'], j['ms_per_step'], j['roofline']['all_convs'])"
python bench.py --steps 8 --warmup 3 $COMMON --range-check lazy 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lazy', j['value'], j['ms_per_step'], j['roofline']['all_convs'])"
python tools/two_stream_probe.py --steps 8 --ways 2 2>/dev/null
cd /tmp
rocprofv3 --kernel-trace -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 $COMMON > /dev/null 2> $O/kt.err
python $GRAFT_REPO_ROOT/tools/rocpd_trace.py /tmp/kt2 4 2>/dev/null > $O/trace_streams2.txt
grep "^# " $O/trace_streams2.txt | tail -12
grep -v "^#" $O/trace_streams2.txt | head -60
