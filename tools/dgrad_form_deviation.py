"""Gradients of one NLL step with the dense blocks' gather data gradients on the Winograd kernels, on the direct scaled kernels and on the
exact fp32-MFMA kernels: per-tensor max-relative deviations (profiles/r06_notes.md section 6).
    python tools/dgrad_form_deviation.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from hcflow_amd import HCFlowNet_SR
from hcflow_amd.config import preset
from tests.util import cached_params, spec_grads
cfg = preset("SR_4X_tiny")
g = torch.Generator().manual_seed(29)
hr = torch.rand(3, 3, 96, 160, generator=g).cuda()
lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
noise = torch.rand(hr.shape, generator=g).cuda()
res = {}
for form, minpix in (("wino", "0"), ("direct", "1000000000"), ("exact", "1000000000")):
    os.environ["HCF_DGRAD_WINO_MIN_PIX"] = minpix
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train().set_precision("exact" if form == "exact" else "f16x3")
    _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
    nll.backward()
    res[form] = spec_grads(net, cfg)
for a_, b_ in (("wino", "exact"), ("direct", "exact"), ("wino", "direct")):
    rel = []
    for a, b in zip(res[a_], res[b_]):
        rel.append(float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30))
    gm = max(float(np.abs(b).max()) for b in res[b_])
    relg = max(float(np.abs(a - b).max()) for a, b in zip(res[a_], res[b_])) / gm
    print("%s vs %s: worst per-tensor max-relative deviation %.2e, median %.2e, relative to the largest gradient %.2e" % (a_, b_, max(rel), float(np.median(rel)), relg))
