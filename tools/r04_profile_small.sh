#!/bin/bash
# Round 4 diagnostic: kernel-level traces of the configurations that are NOT the headline (B = 1 latency, Face x8, training step).
# Run on the GPU box through gpurun; summaries land in gpurun_out/r04_small/.
set -x
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_small
mkdir -p $O
cd $GRAFT_REPO_ROOT
COMMON="--no-other-precision --no-cpu-baseline --no-exact-check"
python bench.py --steps 10 --warmup 3 > $O/bench_b16.json 2> $O/bench_b16.err
python bench.py --batch 1 --steps 30 --warmup 5 $COMMON > $O/bench_b1.json 2>> $O/bench_b1.err
python bench.py --preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 30 --warmup 5 $COMMON > $O/bench_c3.json 2>> $O/bench_c3.err
python bench.py --preset Rescaling_DF2K_4X --batch 8 --lr-size 160 --steps 10 --warmup 3 $COMMON > $O/bench_c4.json 2>> $O/bench_c4.err
python tools/train_bench.py --steps 5 > $O/train.txt 2>&1
for cfg in "b1:--batch 1 --steps 5 --warmup 2" "c3:--preset SR_CelebA_8X --batch 32 --lr-size 20 --steps 5 --warmup 2"; do
  name=${cfg%%:*}; a=${cfg#*:}
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python bench.py $a $COMMON > /dev/null 2> $O/prof_$name.err
  find /tmp/prof_$name -name "*kernel_stats.csv" -exec cp {} $O/kstats_$name.csv \;
  find /tmp/prof_$name -name "*kernel_trace.csv" -exec sh -c 'python - "$1" "$2" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step: the final 1/7 of the rows (5 timed + 2 warmup steps)
n=len(rows)//7
with open(sys.argv[2],"w") as f:
    prev=None
    for r in rows[-n:]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        f.write("%8.2f %8.2f %6s %s\n"%((e-s)/1e3,((s-prev)/1e3 if prev else 0),r.get("Grid_Size_X",r.get("Grid_Size","")),r["Kernel_Name"][:110]))
        prev=e
PY' _ {} $O/trace_$name.txt \;
done
rocprofv3 --kernel-trace --stats -d /tmp/prof_train -o train -- python tools/train_bench.py --steps 3 > /dev/null 2> $O/prof_train.err
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} $O/kstats_train.csv \;
ls -la $O
