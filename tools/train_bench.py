#!/usr/bin/env python
"""Config 5 of BASELINE.json on one GPU: NLL training step of the SR x4 net (HR 160x160 patches, B = 16 per GPU):
forward + backward + Adam, as HCFlow_SR_model.optimize_parameters runs it (:184-205).
    python tools/train_bench.py [--steps 5] [--batch 16] [--hr 160] [--preset SR_DF2K_4X]
Prints per-phase milliseconds (host wall clock around synchronised phases)."""
import argparse
import contextlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--hr", type=int, default=160)
    ap.add_argument("--preset", default="SR_DF2K_4X")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "exact"])
    ap.add_argument("--optim", default="torch", choices=["torch", "native", "torch-fused"],
                    help="torch: torch.optim.Adam + torch clip_grad_norm_ as the reference's caller; native: hcflow_amd.optim")
    args = ap.parse_args()
    cfg = preset(args.preset)
    with contextlib.redirect_stdout(sys.stderr):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().train().set_precision(args.precision)
    g = torch.Generator().manual_seed(0)
    B, H = args.batch, args.hr
    hr = torch.rand(B, 3, H, H, generator=g).cuda()
    lr = torch.nn.functional.interpolate(hr, scale_factor=1.0 / cfg.scale, mode="bicubic", align_corners=False).clamp(0, 1)
    ps = [p for p in net.parameters() if p.requires_grad]
    if args.optim == "native":
        from hcflow_amd import optim as hopt
        opt, clip = hopt.Adam(ps, lr=2.5e-4, betas=(0.9, 0.99)), hopt.clip_grad_norm_
    else:
        opt = torch.optim.Adam(ps, lr=2.5e-4, betas=(0.9, 0.99), **({"fused": True} if args.optim == "torch-fused" else {}))
        clip = torch.nn.utils.clip_grad_norm_
    sync = torch.cuda.synchronize
    rows = []
    for it in range(args.steps + 1):
        sync(); t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        _, nll = net(hr=hr, lr=lr, reverse=False)
        sync(); t1 = time.perf_counter()
        nll.backward()
        sync(); t2 = time.perf_counter()
        clip(net.parameters(), 100.0)
        opt.step()
        sync(); t3 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), float(nll.detach())))
        print("step %d: forward (incl. weight repack) %.1f ms  backward %.1f ms  clip+Adam %.1f ms  nll %.4f" % ((it,) + rows[-1]), flush=True)
    r = rows[1:]
    n = len(r)
    f, b, o = (sum(x[i] for x in r) / n for i in range(3))
    eng = net.engine()
    print("mean over %d steps: forward %.1f  backward %.1f  optimiser %.1f  total %.1f ms  -> %.2f samples/s; "
          "activation arena %.2f GB" % (n, f, b, o, f + b + o, B / ((f + b + o) / 1e3), eng.workspace_bytes() / 2 ** 30))


if __name__ == "__main__":
    main()
