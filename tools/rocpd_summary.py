#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace the way `--stats` would:
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
"""
import glob
import os
import sqlite3
import sys


def main(path):
    if os.path.isdir(path):                      # a rocprofv3 -d directory: take the (first) results db below it
        dbs = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
        if not dbs:
            sys.exit("no *_results.db under %s" % path)
        path = dbs[0]
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                            "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    span = list(cur.execute("select min(start), max(end) from kernels"))[0]
    print("# source: %s" % path)
    print("# total kernel time %.3f ms over a %.3f s span, %d dispatches" % (tot / 1e6, (span[1] - span[0]) / 1e9,
                                                                           sum(r[1] for r in rows)))
    print("%-78s %7s %12s %11s %11s %11s %6s %5s %5s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us",
                                                         "pct", "vgpr", "sgpr", "lds"))
    for r in rows:
        print("%-78s %7d %12.3f %11.1f %11.1f %11.1f %6.2f %5d %5d %6d" % (r[0][:78], r[1], r[2] / 1e6, r[3] / 1e3,
                                                                      r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot,
                                                                      r[6] or 0, r[7] or 0, r[8] or 0))


if __name__ == "__main__":
    main(sys.argv[1])
