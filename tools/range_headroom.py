#!/usr/bin/env python
"""f16x3 range headroom of the conv inputs (profiles/rNN_range_headroom.json).   GPU box:  python tools/range_headroom.py

The f16x3 kernels split every conv input x into f16 hi / lo parts, so |x| (and, in the Winograd kernels, |B^T d B| of the 4x4
input patches, <= 4 max|x|) must stay below 65504; beyond it the pass is re-run on the exact fp32 kernels (3x slower). This
tool runs the engine's range probe (hcf_debug_range_probe: a max-reduction beside every conv launch) over
  (a) the shipped full-depth nets with ActNorms fitted by the REFERENCE's data-dependent initialisation on the reference's own
      example images (tests/golden/net_*_real.npz) -- unit-variance activations, the regime of a trained net -- inverse pass at
      tau 0.8 / 1.0 and forward (NLL / encode) pass;
  (b) the seeded weight recipe (hcflow_amd/params.py) with every conv weight scaled x1 / x2 / x4 / x8 on BASELINE config 2's LR
      size, to show where the guard starts to fire,
and reports per net the largest values, the layers that hold them and the headroom factor 65504 / max."""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling, preset, make_params, eps_shapes  # noqa: E402
from hcflow_amd.config import param_spec  # noqa: E402
from tests.util import load_golden, real_inputs, real_params, seeded_eps  # noqa: E402

F16_MAX = 65504.0


def module(cfg, p):
    with contextlib.redirect_stdout(sys.stderr):         # the constructor prints the reference's `shapes:` line
        net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    return net.to("cuda:0").eval().set_precision("f16x3")


def summarise(recs):
    f16 = [r for r in recs if r[7]]
    xs = sorted(f16, key=lambda r: -r[1])
    vs = sorted([r for r in f16 if r[8]], key=lambda r: -r[2])
    mx, mv = (xs[0][1] if xs else 0.0), (vs[0][2] if vs else 0.0)
    worst = max(mx, mv, 1e-30)
    return {"conv_launches_probed": len(recs), "on_f16x3": len(f16), "winograd": len(vs),
            "max_abs_input": mx, "max_abs_winograd_V": mv, "headroom_factor": F16_MAX / worst,
            "median_max_abs_input": float(np.median([r[1] for r in f16])) if f16 else 0.0,
            "top_inputs": [{"layer": r[0], "max_abs": r[1], "cin": r[3], "cout": r[4], "HxW": [r[5], r[6]]} for r in xs[:5]],
            "top_winograd_V": [{"layer": r[0], "max_abs_V": r[2], "max_abs_input": r[1]} for r in vs[:5]]}


def probe(net, fn):
    eng = net.engine()
    eng.range_probe(True)
    n0 = eng.fallback_count()
    with torch.no_grad():
        fn()
    torch.cuda.synchronize()
    recs = eng.range_probe_records()
    eng.range_probe(False)
    out = summarise(recs)
    out["range_fallbacks"] = eng.fallback_count() - n0
    return out


def main():
    res = {"f16_max": F16_MAX, "note": __doc__.split("\n\n")[1].replace("\n", " "), "reference_fitted": {}, "recipe_scaled": {}}
    for name in ("net_sr4_real", "net_sr8_real", "net_rescale_real"):
        g = load_golden(name)
        cfg, p = real_params(g)
        lr, hr = real_inputs(g)
        B, _, h, w = lr.shape
        net = module(cfg, p)
        tau = float(g["inv1_tau"])
        eps = seeded_eps(cfg, B, h, w, tau, int(g["inv1_eps_seed"]))
        ent = {"weights": "seeded recipe + ActNorms fitted by the reference on these images (tests/golden/%s.npz)" % name,
               "images": str(g["images"]), "lr_shape": list(lr.shape),
               "inverse_tau_%.1f" % tau: probe(net, lambda: net(lr=lr.cuda(), eps_std=tau, reverse=True, eps=eps))}
        if cfg.sr:
            ent["forward_nll"] = probe(net, lambda: net(hr=hr.cuda(), lr=lr.cuda(), reverse=False))
        else:
            ent["forward_encode"] = probe(net, lambda: net(hr=hr.cuda(), reverse=False))
        res["reference_fitted"][name] = ent
        print(name, {k: (v["max_abs_input"], v["max_abs_winograd_V"], v["range_fallbacks"]) for k, v in ent.items() if isinstance(v, dict)},
              file=sys.stderr)
        del net
    cfg = preset("SR_DF2K_4X")
    base = make_params(cfg, 1234)
    kinds = {k: kind for k, _, kind in param_spec(cfg)}
    g = torch.Generator().manual_seed(5)
    lr = torch.rand(2, 3, 160, 160, generator=g).cuda()
    for s in (1.0, 2.0, 4.0, 8.0):
        p = {k: (v * s if kinds[k] in ("conv_w", "fcn_w", "zeros_w") else v.clone()) for k, v in base.items()}
        net = module(cfg, p)
        ent = probe(net, lambda: net(lr=lr, eps_std=0.8, reverse=True, seed=7))
        res["recipe_scaled"]["x%g" % s] = ent
        print("recipe x%g" % s, ent["max_abs_input"], ent["max_abs_winograd_V"], ent["range_fallbacks"], file=sys.stderr)
        del net
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
