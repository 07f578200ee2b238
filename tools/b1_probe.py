#!/usr/bin/env python
"""Why does the B = 1 call (BASELINE config 1) read 13.6 ms in some processes and 22-26 ms in others?  One condition per process:
   python tools/b1_probe.py COND     COND in: fresh | after_b16 | after_b16_s2 | side_stream | lazy | after_b16_side
Prints ms per call (sync policy as the module default unless `lazy`), the host time of one call's enqueue (lazy policy), and the
shader clock sampled during the loop."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    cond = sys.argv[1]
    from hcflow_amd import HCFlowNet_SR, preset, make_params
    import bench
    cfg = preset("SR_DF2K_4X")
    params = make_params(cfg, 1)

    def build():
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(params, strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        return net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        if cond.startswith("after_b16"):
            big = build()
            if cond.startswith("after_b16_s2"):
                big.set_streams(2)
            lr16 = torch.rand(16, 3, 160, 160, generator=g).cuda()
            ctx = torch.cuda.stream(torch.cuda.Stream()) if cond == "after_b16_side" else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                for i in range(int(os.environ.get("B16_STEPS", "8"))):
                    big(lr=lr16, eps_std=0.8, reverse=True, seed=i)
            torch.cuda.synchronize()
            if not cond.endswith("_keep"):
                del big, lr16
                torch.cuda.empty_cache()
        net = build()
        if cond == "lazy":
            net.set_range_check("lazy")
        lr = torch.rand(1, 3, 160, 160, generator=g).cuda()
        side = torch.cuda.Stream() if cond == "side_stream" else None

        def run(n):
            if side is not None:
                with torch.cuda.stream(side):
                    for i in range(n):
                        net(lr=lr, eps_std=0.0, reverse=True)
            else:
                for i in range(n):
                    net(lr=lr, eps_std=0.0, reverse=True)
            torch.cuda.synchronize()
        tw = time.perf_counter()
        while time.perf_counter() - tw < 1.5:
            run(8)
        for rep in range(3):
            with bench.PowerSampler(0) as ps:
                t0 = time.perf_counter()
                run(40)
                dt = (time.perf_counter() - t0) / 40
            pw = ps.block() or {}
            print("%-14s rep %d: %.2f ms per call, sclk %s MHz, %s W" % (cond, rep, dt * 1e3, pw.get("sclk_MHz_avg"), pw.get("avg_W")), flush=True)
        # kernel time of one call (HIP events around every conv launch) against the call's wall time: slower kernels or longer gaps?
        eng = net.engines()[0]
        eng.profile_convs(True)
        t0 = time.perf_counter()
        run(10)
        dt = (time.perf_counter() - t0) / 10
        eng.profile_convs(False)
        ms, n, _, _ = eng.conv_time(0, 0, reset=True)
        print("%-14s with conv events: %.2f ms per call, conv kernels %.2f ms in %d launches per call" % (cond, dt * 1e3, ms / 10, n // 10), flush=True)
        # host enqueue time of one call without the read-back
        net.set_range_check("lazy")
        run(4)
        t0 = time.perf_counter()
        net(lr=lr, eps_std=0.0, reverse=True)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        tt = time.perf_counter() - t0
        print("%-14s host enqueue of one call %.2f ms, until done %.2f ms" % (cond, th * 1e3, tt * 1e3), flush=True)


if __name__ == "__main__":
    main()
