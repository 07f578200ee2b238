"""Is HIP-graph replay of the cached plan worth building into the engine? Eager module calls (lazy policy) against a
torch.cuda.CUDAGraph replay of the same call, per configuration (GPU box). profiles/r03_notes.md section 7."""
import contextlib
import sys
import time

import torch

sys.path.insert(0, '.')
from hcflow_amd import HCFlowNet_SR, preset, make_params

for name, B, h in (("SR_DF2K_4X", 1, 160), ("SR_DF2K_4X", 16, 160), ("SR_CelebA_8X", 32, 20), ("SR_CelebA_8X", 1, 20)):
    cfg = preset(name)
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval().set_range_check("lazy")
    lr = torch.rand(B, 3, h, h).cuda()
    n = 20 if B * h * h < 100000 else 8
    with torch.no_grad():
        for _ in range(3):
            net(lr=lr, eps_std=0.8, reverse=True, seed=7)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            net(lr=lr, eps_std=0.8, reverse=True, seed=7)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(lr=lr, eps_std=0.8, reverse=True, seed=7)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            out = net(lr=lr, eps_std=0.8, reverse=True, seed=7)
        graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            graph.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / n * 1e3
        # sync-per-call variants (what the default policy does)
        t0 = time.perf_counter()
        for i in range(n):
            net(lr=lr, eps_std=0.8, reverse=True, seed=7)
            torch.cuda.synchronize()
        eager_sync = (time.perf_counter() - t0) / n * 1e3
        t0 = time.perf_counter()
        for i in range(n):
            graph.replay()
            torch.cuda.synchronize()
        rep_sync = (time.perf_counter() - t0) / n * 1e3
    print("%-14s B=%2d LR %3d: eager %.2f ms  graph replay %.2f ms | with a sync per call: eager %.2f  replay %.2f" % (name, B, h, eager, rep, eager_sync, rep_sync))
    del net, graph
