"""Per-launch trace of the LAST `1/parts` of a rocprofv3 --kernel-trace run (rocpd database): duration, gap to the previous
kernel's end, grid / workgroup size, kernel name -- every kernel, in start order -- followed by a per-kernel summary of that slice.
    rocprofv3 --kernel-trace -d /tmp/kt -- python bench.py --batch 1 --steps 5 --warmup 2 ...
    python tools/rocpd_trace.py /tmp/kt 7 > profiles/r04_trace_b1.txt"""
import collections
import glob
import os
import sqlite3
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True))[0]
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cur = sqlite3.connect(path).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else "grid_size_x"
wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
print("# columns: %s" % ", ".join(cols), file=sys.stderr)
q = "select name, start, end, %s%s%s from kernels order by start" % (gx, (", " + wx) if wx else ", 0", (", " + qcol) if qcol else "")
rows = list(cur.execute(q))
n = len(rows) // parts
rows = rows[-n:]
agg = collections.OrderedDict()
prev = None
t0 = rows[0][1]
print("# %d launches, slice wall %.3f ms" % (len(rows), (rows[-1][2] - t0) / 1e6))
print("#   dur_us   gap_us  blocks  kernel")
for r in rows:
    name, s, e, g = r[0], r[1], r[2], r[3]
    w = r[4] if wx else 0
    short = name.split("(")[0].replace("hcf::", "")[:100]
    blocks = g // w if w else g
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%9.2f %8.2f %7d  %s%s" % ((e - s) / 1e3, gap, blocks, ("s%s +%.1f " % (r[5], (s - t0) / 1e3)) if qcol else "", short))
    a = agg.setdefault(short, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += max(gap, 0.0)
    prev = e
print("# ---- summary of the slice: calls, total us, avg us, total gap-before us")
tot = 0.0
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("# %5d %10.1f %8.2f %9.1f  %s" % (a[0], a[1], a[1] / a[0], a[2], k))
    tot += a[1]
print("# kernel time %.3f ms, gaps %.3f ms" % (tot / 1e3, sum(a[2] for a in agg.values()) / 1e3))
if qcol:                      # per stream / queue: launches, summed kernel time, union of busy intervals
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(r[5], []).append((r[1], r[2]))
    for k, iv in per.items():
        busy, cur_s, cur_e = 0, None, None
        for s_, e_ in sorted(iv):
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += cur_e - cur_s
        print("# %s %s: %d launches, kernel time %.3f ms, busy (union) %.3f ms, first start +%.3f ms, last end +%.3f ms" % (
            qcol, k, len(iv), sum(e_ - s_ for s_, e_ in iv) / 1e6, busy / 1e6, (min(s_ for s_, _ in iv) - t0) / 1e6,
            (max(e_ for _, e_ in iv) - t0) / 1e6))
