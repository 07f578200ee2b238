#!/usr/bin/env python
"""Sample socket power and shader clock while one conv shape runs back to back (is the plateau a power cap?).
    python tools/power_probe.py [--seconds 4] [--precision f16x3]
Reads the amdgpu hwmon files directly (power1_average/power1_input in uW, freq1_input in Hz, power1_cap) and
falls back to `rocm-smi --json` when they are not there. Prints one line per workload: idle, each conv shape.
"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import _lib  # noqa: E402

SHAPES = [
    ("rdb_conv5_L0  64+128->64 @320", 16, 320, 320, [64, 128], 64, 3),
    ("rdb_conv4_L0  64+96->32  @320", 16, 320, 320, [64, 96], 32, 3),
    ("fcn_conv2     64->64 1x1 @320", 16, 320, 320, [64], 64, 1),
]


def hwmon_files():
    out = {}
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for key in ("power1_average", "power1_input", "freq1_input", "power1_cap", "temp1_input"):
            p = os.path.join(d, key)
            if os.path.exists(p) and key not in out:
                out[key] = p
    return out


def read_int(p):
    try:
        with open(p) as f:
            return int(f.read().strip())
    except Exception:
        return None


def smi_sample():
    try:
        t = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True,
                           timeout=10).stdout
        d = json.loads(t)
        card = d[sorted(d)[0]]
        pw = [float(v) for k, v in card.items() if "ower" in k and "(W)" in k]
        ck = [v for k, v in card.items() if "sclk" in k]
        return (pw[0] if pw else None), (ck[0] if ck else None)
    except Exception as e:   # noqa: BLE001
        return None, str(e)


class Sampler(threading.Thread):
    def __init__(self, files):
        super().__init__(daemon=True)
        self.files, self.rows, self.stop = files, [], False

    def run(self):
        while not self.stop:
            if self.files:
                pw = read_int(self.files.get("power1_average") or self.files.get("power1_input", ""))
                fq = read_int(self.files.get("freq1_input", ""))
                self.rows.append((pw / 1e6 if pw else None, fq / 1e6 if fq else None))
                time.sleep(0.05)
            else:
                self.rows.append(smi_sample())
                time.sleep(0.2)


def summarise(tag, rows, extra=""):
    pw = [r[0] for r in rows if isinstance(r[0], (int, float))]
    fq = [r[1] for r in rows if isinstance(r[1], (int, float))]
    tail = lambda v: v[len(v) // 2:]                                       # noqa: E731  (steady state: second half)
    ps = "power %6.0f W (max %6.0f)" % (sum(tail(pw)) / len(tail(pw)), max(pw)) if pw else "power n/a"
    fs = "sclk %5.0f MHz" % (sum(tail(fq)) / len(tail(fq))) if fq else "sclk %s" % (rows[-1][1] if rows else "n/a")
    print("%-34s %s  %s  samples %d  %s" % (tag, ps, fs, len(rows), extra), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--precision", default="f16x3", choices=["exact", "f16x3"])
    args = ap.parse_args()
    files = hwmon_files()
    print("hwmon:", {k: v for k, v in files.items()}, flush=True)
    if "power1_cap" in files:
        print("power cap: %.0f W" % (read_int(files["power1_cap"]) / 1e6))
    lib = _lib.load()
    assert lib.hcf_op_set_precision(_lib.Engine.PRECISIONS[args.precision]) == 0
    torch.cuda.init()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    s = Sampler(files)
    s.start()
    time.sleep(1.5)
    s.stop = True
    s.join()
    summarise("idle", s.rows)

    for name, B, H, W, srcs, cout, k in SHAPES:
        arr = (C.c_int32 * len(srcs))(*srcs)
        ms, fl = C.c_double(), C.c_double()
        s = Sampler(files)
        s.start()
        t0 = time.time()
        tfs, clks = [], []
        while time.time() - t0 < args.seconds:
            rc = lib.hcf_bench_conv(B, H, W, arr, len(srcs), cout, k, 50, C.byref(ms), C.byref(fl), st)
            assert rc == 0, rc
            tfs.append(fl.value / ms.value / 1e9)
            clks.append(lib.hcf_debug_last_clock_mhz())
        s.stop = True
        s.join()
        summarise(name, s.rows, "%.1f TFLOP/s-eq, in-kernel clk %.0f MHz" % (tfs[-1], clks[-1]))


if __name__ == "__main__":
    main()
