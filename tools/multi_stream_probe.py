#!/usr/bin/env python
"""Probe (MALL residency by SMALL pieces on several streams): W module instances, each on its own HIP stream, each running the
inverse pass on pieces of b samples, free-running (one host thread enqueues the streams in turn, `lazy` range policy, one
synchronisation at the end). A piece's dense-block working set (b x 79 MB at 320^2) times W can sit inside the 256 MB Infinity
Cache, which a B = 16 launch (1.3 GB) cannot; the concurrent streams fill each other's ragged rounds.
    python tools/multi_stream_probe.py            # sweeps (W, b)"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def build(cfg, params):
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(params, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    return net.cuda().eval().set_precision("f16x3").set_range_check("lazy")


def main():
    cfg = preset("SR_DF2K_4X")
    params = make_params(cfg, 1234)
    WMAX = int(os.environ.get("WMAX", "4"))
    nets = [build(cfg, params) for _ in range(WMAX)]
    streams = [torch.cuda.Stream() for _ in range(WMAX)]
    total = int(os.environ.get("IMAGES", "96"))
    lr = torch.rand(16, 3, 160, 160).cuda()
    combos = [(1, 16), (2, 8), (2, 4), (2, 2), (3, 4), (3, 2), (3, 1), (4, 4), (4, 2), (4, 1), (1, 16)]
    with torch.no_grad():
        for W, b in combos:
            if W > WMAX:
                continue
            pieces = max(W, total // b)

            def run(npieces):
                for k in range(npieces):
                    i = k % W
                    with torch.cuda.stream(streams[i]):
                        nets[i](lr=lr[:b], eps_std=0.8, reverse=True, seed=k)
            run(2 * W)                                # plans / workspaces of this piece size
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(pieces)
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("streams %d x pieces of %2d: %6.1f HR img/s  (%d pieces, %.1f ms; host enqueue %.1f ms)"
                  % (W, b, pieces * b / dt, pieces, dt * 1e3, th * 1e3), flush=True)


if __name__ == "__main__":
    main()
