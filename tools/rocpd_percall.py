"""Per-launch durations (start order) of the conv / flow kernels of the LAST third of a rocprofv3 --kernel-trace run:
    rocprofv3 --kernel-trace -d /tmp/kt -- python bench.py --steps 1 --warmup 2 ...; python tools/rocpd_percall.py /tmp/kt
-> profiles/r03_percall_trace.txt"""
import glob, os, sqlite3, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))[0]
cur = sqlite3.connect(path).cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'kernel' in t.lower()][:20])
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
# take the last third of the run (timed step)
n=len(rows)
import collections
agg=collections.OrderedDict()
seq=[]
for name,s,e,g in rows[-(n//3):]:
    short = name.split('(')[0][:90]
    seq.append((short,(e-s)/1e3,g))
for s in seq:
    if 'conv_f16x3_kernel<2, true, false, true' in s[0] or 'false, false, 8' in s[0] or 'false, false, 12' in s[0] or 'false, false, 24' in s[0] or 'step_tail' in s[0] or 'gauss' in s[0] or 'wino' in s[0].lower():
        print("%-95s %8.1f us grid %d"%s)
