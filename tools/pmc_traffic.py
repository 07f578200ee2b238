#!/usr/bin/env python
"""HBM bytes per launch of the conv kernels from two rocprofv3 counter passes (MI355X_MICROARCH.md, HBM section:
FETCH_SIZE and WRITE_SIZE do not fit one pass; on gfx950 FETCH_SIZE counts 64 B per 128-B request -> doubled;
both counters are in KiB... see UNIT below).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-check
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ... (same)
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_traffic_pmc.json
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

UNIT = 1024.0          # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
FETCH_FIX = 2.0        # gfx950: 128-B read requests are tallied as 64 B


def short_name(k):
    m = re.search(r"conv_f16x3_kernel<(\d), (true|false), (true|false), (true|false), (\d+), (\d+)(?:, (?:true|false))*>", k)
    if m:
        ntb, vec, up, fuse2, tailc, th = m.groups()
        tag = "f16x3<%s>" % ntb
        if fuse2 == "true":
            tag += "+fuse2"
        if tailc != "0":
            tag += "+tail%s" % tailc
        if up == "true":
            tag += "+up"
        return tag
    m = re.search(r"fcn12_kernel<(true|false)>", k)
    if m:
        return "fcn12" + ("+pre" if m.group(1) == "true" else "")
    m = re.search(r"conv_wino4_kernel<(\d)(?:, false)?>", k)      # (<RES, true>: the training pass' scaled variants, not part of these tables)
    if m:
        return "wino4<%s>" % m.group(1)         # the 64-output-channel Winograd kernel (RDB conv5: 1 or 2 residual inputs)
    m = re.search(r"conv_wino2_kernel<(\d)(?:, false)?>", k)
    if m:
        return "wino<%s>" % m.group(1)          # template argument = number of residual inputs (0: conv3 / conv4, 1 / 2: conv5)
    m = re.search(r"conv_mfma_kernel<(\d), (\d), (true|false)>", k)
    if m:
        return "exact<%s,%s>" % (m.group(1), m.group(2))
    return None


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit("no *counter_collection.csv under %s" % d)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            key = short_name(row["Kernel_Name"])
            if key:
                acc[key][0] += float(row["Counter_Value"])
                acc[key][1] += 1
    return acc


def main(fetch_dir, write_dir):
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 5 --warmup 2 --no-other-precision "
                     "--warmup 1`, B=16; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 64 B per 128 B "
                     "request); averaged over all launches of the kernel in the run (tools/pmc_traffic.py)",
           "kernels": {}}
    for k in sorted(set(fe) & set(wr)):
        f = fe[k][0] / fe[k][1] * UNIT * FETCH_FIX
        w = wr[k][0] / wr[k][1] * UNIT
        out["kernels"][k] = {"launches_sampled": fe[k][1], "fetch_bytes_per_launch": f, "write_bytes_per_launch": w,
                             "hbm_bytes_per_launch": f + w}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
