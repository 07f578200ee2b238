#!/usr/bin/env python
"""Probe: why is a B = 1 call on the process' null stream ~2x slower after torch side streams have been used in the process?
    python tools/stream_leftover_probe.py"""
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def main():
    cfg = preset("SR_DF2K_4X")
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval()
    lr1 = torch.rand(1, 3, 160, 160).cuda()
    lr8 = torch.rand(8, 3, 160, 160).cuda()

    def lat(tag, stream=None, n=15):
        ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
        with torch.no_grad(), ctx:
            for _ in range(3):
                net(lr=lr1, eps_std=0.0, reverse=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                net(lr=lr1, eps_std=0.0, reverse=True)
            torch.cuda.synchronize()
        print("%-70s %.2f ms" % (tag, 1e3 * (time.perf_counter() - t0) / n), flush=True)

    lat("A  fresh process, null stream")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    lat("B  two side streams created, never used")
    with torch.cuda.stream(s1):
        a = torch.zeros(1024, device="cuda") + 1
    with torch.cuda.stream(s2):
        b = torch.zeros(1024, device="cuda") + 1
    torch.cuda.synchronize()
    lat("C  one tiny kernel ran on each side stream")
    lat("D  the B = 1 calls issued ON a side stream", stream=s1)
    lat("E  null stream again")
    with torch.no_grad():
        net.set_streams(2)
        for _ in range(2):
            net(lr=lr8, eps_std=0.8, reverse=True)
        torch.cuda.synchronize()
        net.set_streams(1)
    lat("F  after two split calls (two engines, helper thread), null stream")
    lat("G  after the split calls, B = 1 on a side stream", stream=s1)
    s3 = torch.cuda.Stream()
    lat("H  B = 1 on a fresh third side stream", stream=s3)


if __name__ == "__main__":
    main()
