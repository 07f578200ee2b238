#!/usr/bin/env python
"""Micro-benchmark of conv_mfma_kernel on the shapes that dominate config 2 (B=16).
    python tools/conv_bench.py [--iters 10]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import _lib  # noqa: E402

SHAPES = [
    # name, B, H, W, srcs, cout, k
    ("rdb_conv5_L0  64+128->64 @320", 16, 320, 320, [64, 128], 64, 3),
    ("rdb_conv4_L0  64+96->32  @320", 16, 320, 320, [64, 96], 32, 3),
    ("rdb_conv1_L0  64->32     @320", 16, 320, 320, [64], 32, 3),
    ("rdb_conv5_L1  64+128->64 @160", 16, 160, 160, [64, 128], 64, 3),
    ("rdb_conv2_L1  64+32->32  @160", 16, 160, 160, [64, 32], 32, 3),
    ("fcn_conv1_L0c 3+128->64  @320", 16, 320, 320, [3, 128], 64, 3),
    ("fcn_conv3_L0c 64->6      @320", 16, 320, 320, [64], 6, 3),
    ("fcn_conv2     64->64 1x1 @320", 16, 320, 320, [64], 64, 1),
    ("completion    32->32     @320", 16, 320, 320, [32], 32, 3),      # the direct kernel on the fat schedule's completion shape
    ("completion    32->32     @160", 16, 160, 160, [32], 32, 3),
    # training sizes (config 5: B = 16, LR 40 x 40): the gather data-gradient shapes of a dense block, forward kernels
    ("train dgrad x1 96+64->32  @40", 16, 40, 40, [96, 64], 32, 3),
    ("train dgrad x4 64->32     @40", 16, 40, 40, [64], 32, 3),
    ("train dgrad x0 128+64->64 @40", 16, 40, 40, [128, 64], 64, 3),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--batch", type=int, default=0, help="override the batch size of every shape (0: as listed)")
    ap.add_argument("--precision", default="exact", choices=["exact", "f16x3"])
    ap.add_argument("--ablate", type=int, default=0,
                    help="experiment switches (hcf_debug_set_ablation): 1 / 2 16-row tile for the 32 / 64-channel kernel, "
                         "64 scalar epilogue. Phase timers / the in-kernel clock need the measurement build: "
                         "make -C hcflow_amd/csrc TIMERS=1 && HCFLOW_LIB=hcflow_amd/libhcflow_hip_timers.so python tools/conv_bench.py")
    args = ap.parse_args()
    lib = _lib.load()
    assert lib.hcf_op_set_precision(_lib.Engine.PRECISIONS[args.precision]) == 0
    peak = 157.3 if args.precision == "exact" else 2500.0 / 3
    assert lib.hcf_debug_set_ablation(args.ablate) == 0
    torch.cuda.init()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, (name, B, H, W, srcs, cout, k) in enumerate(SHAPES):
        if args.only >= 0 and i != args.only:
            continue
        if args.batch:
            B = args.batch
        arr = (C.c_int32 * len(srcs))(*srcs)
        ms, fl = C.c_double(), C.c_double()
        rc = lib.hcf_bench_conv(B, H, W, arr, len(srcs), cout, k, args.iters, C.byref(ms), C.byref(fl), st)
        assert rc == 0, rc
        tf = fl.value / ms.value / 1e9
        print("%-34s %9.1f us  %7.2f TFLOP/s-equiv  (%5.1f %% of %.1f %s)" % (
            name, ms.value * 1e3, tf, 100 * tf / peak, peak, args.precision),
            " clk %.0f MHz" % lib.hcf_debug_last_clock_mhz(), flush=True)


if __name__ == "__main__":
    main()
