#!/usr/bin/env python
"""Latency of the drop-in's real use: test_HCFlow.py feeds ONE image per call (data/__init__.py:24); a DIV2K validation image is
~2040 x 1356, i.e. LR 510 x 339 at x4.   python tools/div2k_latency.py [--h 339 --w 510 --iters 10]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=339)
    ap.add_argument("--w", type=int, default=510)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--preset", default="SR_DF2K_4X")
    args = ap.parse_args()
    from hcflow_amd import HCFlowNet_SR, preset, make_params
    cfg = preset(args.preset)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(0)
    lr = torch.rand(1, 3, args.h, args.w, generator=g).cuda()
    for prec in ("f16x3", "exact"):
        net.set_precision(prec)
        with torch.no_grad():
            for _ in range(2):
                out = net(lr=lr, eps_std=0.8, reverse=True, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.iters):
                out = net(lr=lr, eps_std=0.8, reverse=True, seed=2 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.iters
        print("%s  LR %dx%d -> HR %dx%d  %-6s %.1f ms per image (%.2f images/s), fallbacks %d" % (
            args.preset, args.h, args.w, out.shape[2], out.shape[3], prec, dt * 1e3, 1.0 / dt, net.engine().fallback_count()), flush=True)


if __name__ == "__main__":
    main()
