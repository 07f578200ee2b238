"""Rescaling training step: per-tensor gradient deviation of the HIP path from the oracle's autograd, for the forward
pass alone, the inverse pass alone (fixed quantised LR input, incl. d/d lr) or the whole step.
    python tools/dbg_rescale_grads.py fwd|inv|all"""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcflow_amd import HCFlowNet_Rescaling
from hcflow_amd.config import param_spec
from oracle import hcflow_oracle as O
from tests.util import load_golden, params_for, t
from tests.test_oracle_golden import rgrad_eps, rescale_step_loss
g = load_golden("grad_rescale_tiny")
cfg, p = params_for(g)
net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
net.load_state_dict(p, strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__:
        m.inited = True
net = net.to("cuda:0").train()
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
hr, lr, eps = t(g["hr"]), t(g["lr"]), rgrad_eps(g)
q = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in p.items()}
lrq = ((torch.clamp(t(g["fake_lr"]), 0, 1) * 255.).round() / 255.)


def loss_of(fwd, inv, dev):
    if mode == "fwd":
        fl, z1, z2 = fwd(hr.to(dev))
        return 5e-2 * F.mse_loss(fl, lr.to(dev)) + 1e-5 * (torch.cat([z1.flatten(), z2.flatten()], 0) ** 2).mean(), None
    if mode == "inv":
        x = lrq.to(dev).clone().requires_grad_(True)
        fh = inv(x, [e.to(dev) for e in eps])
        return F.l1_loss(fh, hr.to(dev)), x
    l = rescale_step_loss(fwd, inv, hr.to(dev), lr.to(dev), [e.to(dev) for e in eps])
    return l[0] + l[1] + l[2], None


lo, xo = loss_of(lambda x: O.rescale_forward(x, q, cfg), lambda x, e: O.rescale_inverse(x, q, cfg, 1.0, eps=e), "cpu")
lo.backward()
lg, xg = loss_of(lambda x: net(hr=x, u=None, reverse=False), lambda x, e: net(lr=x, z=None, u=None, eps_std=1.0, reverse=True, eps=e), "cuda:0")
lg.backward()
print(mode, "loss oracle %.8f hip %.8f" % (float(lo.detach()), float(lg.detach())))
if xo is not None:
    d = (xg.grad.cpu() - xo.grad)
    print("d/d lr: rel err %.2e (|g| %.3e)" % (float(d.norm() / xo.grad.norm()), float(xo.grad.norm())))
sd = dict(net.named_parameters())
gmax = max(float(q[k].grad.norm()) for k, _, _ in param_spec(cfg) if torch.is_tensor(q[k]) and q[k].grad is not None)
rows = []
for k, _, _ in param_spec(cfg):
    ref = q[k].grad if (torch.is_tensor(q[k]) and q[k].requires_grad) else None
    if ref is None:
        continue
    have = sd[k].grad.cpu() if sd[k].grad is not None else torch.zeros_like(ref)
    rows.append((float((have - ref).norm()) / max(float(ref.norm()), 2e-5 * gmax), float(ref.norm()) / gmax, k))
rows.sort(reverse=True)
for r in rows[:12]:
    print("rel err %.2e  |g|/gmax %.1e  %s" % r)
print("---- level0 cond step 1")
for k, _, _ in param_spec(cfg):
    if "level0_condFlow.additional_flow_steps.1." in k or "level0_condFlow.additional_flow_steps.0.affine.f.conv1" in k:
        ref = q[k].grad
        have = sd[k].grad.cpu()
        print("%-70s rel %.2e  |g| %.3e" % (k[5:], float((have - ref).norm() / ref.norm()), float(ref.norm())))
k = "flow.level0_condFlow.additional_flow_steps.1.affine.f.conv1.weight"
ref, have = q[k].grad, sd[k].grad.cpu()
e = (have - ref).pow(2).sum(dim=(0, 2, 3)).sqrt() / ref.pow(2).sum(dim=(0, 2, 3)).sqrt().clamp_min(1e-12)
print("conv1.weight rel err per input channel:", [round(float(x), 5) for x in e[:8]], "... max rest", float(e[8:].max()))
e = (have - ref).pow(2).sum(dim=(0, 1)).sqrt() / ref.pow(2).sum(dim=(0, 1)).sqrt()
print("per tap:", [round(float(x), 5) for x in e.flatten()])
k = "flow.level0_condFlow.additional_flow_steps.1.affine.f.conv1.actnorm.bias"
print("bias have/ref first 8:", sd[k].grad.cpu().flatten()[:8].tolist(), q[k].grad.flatten()[:8].tolist())
