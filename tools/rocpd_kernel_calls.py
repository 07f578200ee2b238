#!/usr/bin/env python
"""List the distinct launch shapes of one kernel in a rocprofv3 rocpd database: grid, count, mean / total time.
    python tools/rocpd_kernel_calls.py <dir-or-db> <kernel-name-substring>"""
import glob
import os
import sqlite3
import sys


def main(path, pat):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[0]
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    gx = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
    sel = ", ".join(c for c in ("grid_x", "grid_y", "grid_z") if c in cols) or (gx[0] if gx else "0")
    q = ("select %s, count(*), min(end-start), max(end-start), avg(end-start), sum(end-start) from kernels where name like ? group by %s order by sum(end-start) desc"
         % (sel, sel))
    rows = list(cur.execute(q, ("%" + pat + "%",)))
    print("# columns available:", cols)
    for r in rows[:40]:
        print(r[:-5], "calls %d  min %.1f  max %.1f  avg %.1f us  total %.2f ms" % (
            r[-5], r[-4] / 1e3, r[-3] / 1e3, r[-2] / 1e3, r[-1] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
