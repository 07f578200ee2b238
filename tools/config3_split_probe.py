#!/usr/bin/env python
"""Face x8 (BASELINE config 3: B = 32, LR 20 x 20, tau sweep) with one stream, the two-stream split, and the split with the second
half enqueued by the helper thread -- Python's cyclic GC off inside the timed loops (the earlier A/B numbers of this configuration
were taken with 50-70 ms collection pauses falling into 8-call regions).   python tools/config3_split_probe.py"""
import contextlib
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def main():
    cfg = preset("SR_CelebA_8X")
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval()
    lr = torch.rand(32, 3, 20, 20).cuda()
    taus = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]

    def run(n, k0=0):
        for i in range(n):
            net(lr=lr, eps_std=taus[i % 10], reverse=True, seed=k0 + i)
        torch.cuda.synchronize()
    with torch.no_grad():
        for policy in ("sync", "lazy"):
            net.set_range_check(policy)
            for streams, thr in ((1, None), (2, "0"), (2, "1"), (1, None)):
                net.set_streams(streams)
                if thr is None:
                    os.environ.pop("HCF_SPLIT_THREADED", None)
                else:
                    os.environ["HCF_SPLIT_THREADED"] = thr
                run(30)
                gc.collect()
                gc.disable()
                ts = []
                for rep in range(3):
                    t0 = time.perf_counter()
                    run(40, 100 * rep)
                    ts.append((time.perf_counter() - t0) / 40)
                gc.enable()
                print("%-5s streams %d threaded %-4s: %s ms per call -> %.0f img/s" % (
                    policy, streams, thr, " / ".join("%.2f" % (1e3 * t) for t in ts), 32 / min(ts)), flush=True)


if __name__ == "__main__":
    main()
