"""Every sample of the timed configuration against the CPU oracle (the GPU suite's test_config2_b16_timed_configuration_vs_cpu_oracle
checks samples 0 / 7 / 15 to stay within a minute; this is the same comparison over all 16, kept as evidence under profiles/):
ONE B = 16 engine call at BASELINE config 2 (full depth, LR 160 x 160, tau 0.8, module default f16x3), 16 B = 1 oracle passes.
    python tools/fullsize_parity_all_samples.py      (runs on the GPU box: the oracle is tests' checker, never the product)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hcflow_oracle as O
from hcflow_amd import HCFlowNet_SR
from hcflow_amd.config import preset, eps_shapes
from hcflow_amd.params import make_params

torch.set_num_threads(min(16, os.cpu_count() or 1))
cfg = preset("SR_DF2K_4X")
p = make_params(cfg, 1234)
net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(p, strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__:
        m.inited = True
net = net.to("cuda:0").eval().set_precision("f16x3")
B, tau = 16, 0.8
g = torch.Generator().manual_seed(1616)
lr = torch.rand(B, 3, 160, 160, generator=g)
eps = [torch.randn(s, generator=g) * tau for s in eps_shapes(cfg, B, 160, 160)]
with torch.no_grad():
    n0 = net.engine().fallback_count()
    raw = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=tau, eps=eps, clamp=False).cpu()
    out = net(lr=lr.cuda(), z=None, u=None, eps_std=tau, reverse=True, eps=eps).cpu()
    assert net.engine().fallback_count() == n0
    worst = 0.0
    for b in range(B):
        t0 = time.time()
        ref = O.sr_inverse(lr[b:b + 1], p, cfg, tau, [e[b:b + 1] for e in eps], clamp=False)
        scale = max(1.0, float(ref.abs().max()))
        d_raw = float((raw[b:b + 1] - ref).abs().max())
        d_out = float((out[b:b + 1] - ref.clamp(0, 1)).abs().max())
        worst = max(worst, d_raw / scale)
        print("sample %2d: max |HIP - CPU oracle| unclamped %.2e (scale %.2f), clamped %.2e   [oracle pass %.1f s]" % (b, d_raw, scale, d_out, time.time() - t0),
              flush=True)
        assert d_raw <= 1e-4 * scale and d_out <= 1e-4, b
print("all %d samples of the timed B = 16 f16x3 call within 1e-4 of the CPU oracle: worst relative %.2e" % (B, worst))
