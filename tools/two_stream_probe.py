#!/usr/bin/env python
"""Probe: does running two half-batches concurrently (two engines, two HIP streams, two host threads) beat one
full-batch pass? Kernels of the two streams can fill each other's tail waves and barrier stalls.
    python tools/two_stream_probe.py [--batch 16] [--steps 8]"""
import argparse
import contextlib
import os
import sys
import threading
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
    return net.cuda().eval().set_precision("f16x3")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--ways", type=int, default=2)
    ap.add_argument("--preset", default="SR_DF2K_4X")
    ap.add_argument("--lr-size", type=int, default=160)
    ap.add_argument("--threads", type=int, default=1, help="1: ONE host thread enqueues the streams in turn (what a module would do); 0: a thread per stream")
    args = ap.parse_args()
    cfg = preset(args.preset)
    params = make_params(cfg, 1234)
    nets = [build(cfg, params) for _ in range(args.ways)]
    B = args.batch
    lr = torch.rand(B, 3, args.lr_size, args.lr_size).cuda()
    with torch.no_grad():
        for n in nets:
            n(lr=lr[:B // args.ways], eps_std=0.8, reverse=True)
        nets[0](lr=lr, eps_std=0.8, reverse=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nets[0](lr=lr, eps_std=0.8, reverse=True)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / args.steps
        streams = [torch.cuda.Stream() for _ in range(args.ways)]
        parts = lr.chunk(args.ways)

        def work(i):
            with torch.no_grad(), torch.cuda.stream(streams[i]):
                for _ in range(args.steps):
                    nets[i](lr=parts[i], eps_std=0.8, reverse=True)

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.threads == 0:
            th = [threading.Thread(target=work, args=(i,)) for i in range(args.ways)]
            [t.start() for t in th]
            [t.join() for t in th]
        else:
            for n in nets:
                n.set_range_check("lazy")
            for _ in range(args.steps):
                for i in range(args.ways):
                    with torch.cuda.stream(streams[i]):
                        nets[i](lr=parts[i], eps_std=0.8, reverse=True)
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / args.steps
    print("one stream, B=%d: %.1f ms/step = %.1f img/s;  %d streams x B=%d: %.1f ms/step = %.1f img/s  (%+.1f %%)" % (
        B, 1e3 * one, B / one, args.ways, B // args.ways, 1e3 * two, B / two, 100 * (one / two - 1)))


if __name__ == "__main__":
    main()
