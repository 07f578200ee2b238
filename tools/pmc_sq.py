#!/usr/bin/env python
"""Summarise SQ issue counters of the conv kernels from a rocprofv3 --pmc CSV run:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \\
              SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d OUT -- python tools/conv_bench.py ...
    python tools/pmc_sq.py OUT > profiles/rNN_pmc_sq.json"""
import collections
import csv
import glob
import json
import os
import sys


def main(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_f16x3" not in k and "conv_mfma" not in k and "conv_wgrad" not in k and "conv_wino" not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    out = {}
    for k, c in acc.items():
        m = {n: v / max(cnt[k][n], 1) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        out[k] = {"launches": max(cnt[k].values()), "per_launch": {n: round(v, 1) for n, v in sorted(m.items())},
                  "fractions_of_wave_cycles": {
                      "parked (SQ_WAIT_ANY)": round(m.get("SQ_WAIT_ANY", 0) / wc, 4),
                      "issue-stalled (SQ_WAIT_INST_ANY)": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                      "issuing (SQ_ACTIVE_INST_ANY)": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                      "issuing VALU": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                      "issuing LDS": round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 4)},
                  "mfma_busy_cycles_per_simd_cycle_note": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x shader clock)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1])
