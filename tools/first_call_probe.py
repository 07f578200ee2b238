#!/usr/bin/env python
"""Time to the first output of a batched call (engine builds: host-side packing + upload; the split's twin engine included)."""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402

cfg = preset("SR_DF2K_4X")
params = make_params(cfg, 1)
for streams in (1, 2):
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(params, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval().set_streams(streams)
    lr = torch.rand(16, 3, 160, 160).cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        net(lr=lr, eps_std=0.8, reverse=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    with torch.no_grad():
        net(lr=lr, eps_std=0.8, reverse=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("streams %d: first call %.2f s, second call %.3f s" % (streams, t1 - t0, t2 - t1), flush=True)
    del net
