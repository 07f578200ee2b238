"""A/B of the NLL training step as one batch vs two half batches on two engines / streams (HCFLOW_TRAIN_SPLIT=1, arch.py:
_SRNLLStepSplit): gradient agreement on one step, then free-running step time of both forms in ONE process.
    python tools/train_split_probe.py [--batch 16] [--hr-size 160] [--steps 10]
"""
import argparse, contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hcflow_amd import HCFlowNet_SR, preset, make_params

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--hr-size", type=int, default=160)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = preset("SR_DF2K_4X")
g = torch.Generator().manual_seed(2000)
hr = torch.rand(a.batch, 3, a.hr_size, a.hr_size, generator=g).to(dev)
lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
noise = torch.rand(hr.shape, generator=g).to(dev)
params = make_params(cfg, 1234)
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(params, strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__:
        m.inited = True
net = net.to(dev).train()
ps = [q for q in net.parameters() if q.requires_grad]


def grads(split):
    os.environ["HCFLOW_TRAIN_SPLIT"] = "1" if split else "0"
    for q in ps:
        q.grad = None
    _, nll = net.normal_flow_diracLR(hr, lr, noise=noise)
    nll.backward()
    torch.cuda.synchronize()
    return float(nll), torch.cat([q.grad.flatten() for q in ps]).clone()


n0, g0 = grads(False)
n1, g1 = grads(True)
n0b, g0b = grads(False)
den = float(g0.abs().max())
print("nll one batch %.7f  split %.7f  | max |dg| split vs one %.3e (rel. to max |g| %.3e: %.2e); one vs one again %.3e" %
      (n0, n1, float((g1 - g0).abs().max()), den, float((g1 - g0).abs().max()) / den, float((g0b - g0).abs().max())), flush=True)
rel = ((g1 - g0).norm() / g0.norm()).item()
print("relative L2 difference of the flat gradient: %.3e" % rel, flush=True)

for split in (False, True, False, True):
    os.environ["HCFLOW_TRAIN_SPLIT"] = "1" if split else "0"
    net.load_state_dict(params, strict=True)
    opt = torch.optim.Adam(ps, lr=2.5e-4, betas=(0.9, 0.99))

    def one():
        opt.zero_grad(set_to_none=True)
        _, l = net(hr=hr, lr=lr, reverse=False)
        l.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)
        opt.step()
        return l
    for _ in range(3):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        l = one()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("split %d: %.2f ms per step (%d steps, B = %d), nll %.5f" % (split, 1e3 * dt / a.steps, a.steps, a.batch, float(l)), flush=True)
