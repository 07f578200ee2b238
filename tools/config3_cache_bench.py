#!/usr/bin/env python
"""Config 3 of BASELINE.json (Face x8, B = 32, LR 20x20 -> 160x160) as a tau / sample sweep over ONE LR batch, with and
without the conditional-feature cache (cache_cond=True: HCF_FLAG_KEEP_COND / REUSE_COND, FlowNet_SR_x8.py:129).
    python tools/config3_cache_bench.py [--sweeps 20] > profiles/rNN_config3_cache.json"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from hcflow_amd import HCFlowNet_SR, preset, make_params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--preset", default="SR_CelebA_8X")
    ap.add_argument("--lr-size", type=int, default=20)
    ap.add_argument("--policies", default="sync,lazy", help="range-check policies to time (sync = the module default)")
    args = ap.parse_args()
    cfg = preset(args.preset)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 1234), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval()
    g = torch.Generator().manual_seed(3)
    lr = torch.rand(args.batch, 3, args.lr_size, args.lr_size, generator=g).cuda()
    taus = [0.0, 0.2, 0.4, 0.6, 0.8, 0.85, 0.9, 0.95, 1.0]
    all_res = {}
    outs = {}
    default_policy = net._range_check[0]
    with torch.no_grad():
      for policy in args.policies.split(","):
        net.set_range_check(policy)
        res = all_res.setdefault("range_check=" + policy, {})
        for cached in (False, True):
              for it in range(3):                                    # warm-up (plans, first KEEP call)
                  net(lr=lr, eps_std=0.8, reverse=True, seed=7, cache_cond=cached)
              torch.cuda.synchronize()
              e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
              e0.record()
              n = 0
              for s in range(args.sweeps):
                  for tau in taus:
                      o = net(lr=lr, eps_std=tau, reverse=True, seed=100 + s, cache_cond=cached)
                      n += 1
              e1.record()
              torch.cuda.synchronize()
              ms = e0.elapsed_time(e1) / n
              res["cached" if cached else "uncached"] = {"ms_per_call": round(ms, 3), "img_per_s": round(args.batch / ms * 1e3, 1)}
              outs[cached] = o
        res["bit_identical_last_call"] = bool(torch.equal(outs[False], outs[True]))
    res = all_res
    res["default_policy"] = default_policy
    res["config"] = {"preset": args.preset, "batch": args.batch, "lr": args.lr_size, "calls_timed": args.sweeps * len(taus),
                     "note": "same LR batch, tau sweep x seeds; precision = module default (f16x3)"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
