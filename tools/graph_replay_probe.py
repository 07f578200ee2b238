#!/usr/bin/env python
"""VERDICT r04 task 6: does replaying a steady-state call as ONE HIP graph pay at the small-grid configurations?

Config 1 (SR x4, B = 1, LR 160x160, tau 0) and config 3 (Face x8, B = 32, LR 20x20, one tau): the same call eager (the module's
default "sync" range policy, and "lazy" = what a capture needs) and as torch.cuda.CUDAGraph replays (same kernels, same arguments:
the capture freezes the eps seed, so the probe is a TIMING probe; config 1 draws nothing at tau 0).
"""
import contextlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def build(name, dev):
    cfg = preset(name)
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    return net.to(dev).eval().set_precision("f16x3")


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def probe(name, lr, tau, steps):
    dev = lr.device
    net = build(name, dev)
    res = {}
    with torch.no_grad():
        call = lambda: net(lr=lr, z=None, u=None, eps_std=tau, reverse=True, seed=5)   # noqa: E731
        net.set_range_check("sync")
        res["eager_sync_ms"] = round(timed(call, steps), 3)
        net.set_range_check("lazy")
        res["eager_lazy_ms"] = round(timed(call, steps), 3)
        ref = call()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            call()
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            out = call()
        graph.replay()
        torch.cuda.synchronize()
        res["replay_bit_identical"] = bool(torch.equal(out, ref))
        res["graph_replay_ms"] = round(timed(graph.replay, steps), 3)
        res["range_flag_after"] = bool(net.check_range())
    return res


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(77)
    out = {"config1 SR x4 B=1 LR 160x160 tau 0": probe("SR_DF2K_4X", torch.rand(1, 3, 160, 160, generator=g).to(dev), 0.0, 40),
           "config3 Face x8 B=32 LR 20x20 tau 0.8": probe("SR_CelebA_8X", torch.rand(32, 3, 20, 20, generator=g).to(dev), 0.8, 40),
           "config2 SR x4 B=16 LR 160x160 tau 0.8": probe("SR_DF2K_4X", torch.rand(16, 3, 160, 160, generator=g).to(dev), 0.8, 8)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
