"""Numerical justification of the f16x3 split-precision convolutions: emulates the split
in the CPU oracle and compares the full-depth nets against an fp64 evaluation.
    python tools/split_precision_check.py
"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import hcflow_oracle as O
from hcflow_amd.config import preset, eps_shapes
from hcflow_amd.params import make_params

S = 2048.0
def split(t):
    hi = t.half().float()
    lo = ((t - hi) * S).half().float() / S
    return hi, lo

MODE = {"m": "exact"}
_orig = F.conv2d
def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    w = w.to(x.dtype)
    if MODE["m"] == "exact":
        return _orig(x, w, b, stride, padding, *a, **k)
    if MODE["m"] == "f16x3":
        xh, xl = split(x); wh, wl = split(w)
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) + _orig(xl, wh, None, stride, padding))
    elif MODE["m"] == "f16x3u":     # what hcf_conv_f16x3.hip does: a_lo UNSCALED (f16 subnormals allowed), b planes * 2^11
        xh = x.half().float(); xl = (x - xh).half().float()
        wh, wl = split(w)
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) + _orig(xl, wh, None, stride, padding))
    elif MODE["m"] == "bf16x3":
        xh = x.bfloat16().float(); xl = (x - xh).bfloat16().float()
        wh = w.bfloat16().float(); wl = (w - wh).bfloat16().float()
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) + _orig(xl, wh, None, stride, padding))
    elif MODE["m"] in ("f16x2_no_alo", "f16x2_no_wlo"):     # two of the three products (round 6: is the third one needed? yes: 6-8e-3)
        xh = x.half().float(); xl = (x - xh).half().float()
        wh, wl = split(w)
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) if MODE["m"] == "f16x2_no_alo" else _orig(xl, wh, None, stride, padding))
    elif MODE["m"] == "f16":
        y = _orig(x.half().float(), w.half().float(), None, stride, padding)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y
F.conv2d = conv2d

torch.set_num_threads(8)
for name, h in (("SR_DF2K_4X", 24), ("SR_CelebA_8X", 10), ("Rescaling_DF2K_4X", 24)):
    cfg = preset(name); p = make_params(cfg, 1234)
    p64 = {k: v.double() for k, v in p.items()}
    g = torch.Generator().manual_seed(0)
    lr = torch.rand(2, 3, h, h, generator=g)
    eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 2, h, h)]
    inv = O.sr_inverse if cfg.sr else O.rescale_inverse
    with torch.no_grad():
        MODE["m"] = "exact"
        ref64 = inv(lr.double(), p64, cfg, 0.8, [e.double() for e in eps], clamp=False)
        ref32 = inv(lr, p, cfg, 0.8, eps, clamp=False)
        out = {}
        for m in ("f16x3", "f16x3u", "bf16x3", "f16", "f16x2_no_alo", "f16x2_no_wlo"):
            MODE["m"] = m
            out[m] = inv(lr, p, cfg, 0.8, eps, clamp=False)
    sc = float(ref64.abs().max())
    print(name, "scale %.2f" % sc, "fp32 vs fp64 %.2e" % float((ref32.double()-ref64).abs().max()),
          " ".join("%s vs fp64 %.2e (vs fp32 %.2e)" % (m, float((o.double()-ref64).abs().max()), float((o-ref32).abs().max())) for m, o in out.items()), flush=True)
