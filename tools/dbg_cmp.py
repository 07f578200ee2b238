import sys, os, torch
sys.path.insert(0, os.getcwd())
from hcflow_amd import HCFlowNet_SR, preset, make_params, eps_shapes
name, B, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cfg = preset(name); p = make_params(cfg, 1234)
net = HCFlowNet_SR(opt=cfg.to_opt(), step=0); net.load_state_dict(p, strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__: m.inited = True
net = net.cuda().eval()
g = torch.Generator().manual_seed(5)
lr = torch.rand(B, 3, h, h, generator=g).cuda()
eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, h, h)]
with torch.no_grad():
    ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
    net.set_precision("f16x3")
    fa = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
d = (fa - ex).abs()
print(name, B, h, "env", {k: v for k, v in os.environ.items() if k.startswith("HCF_")}, "max diff %.3e" % float(d.max()), "scale %.2f" % float(ex.abs().max()),
      "bad frac %.2e" % float((d > 1e-3).float().mean()), "per-sample max", [round(float(d[i].max()), 6) for i in range(B)], "fallbacks", net.engine().fallback_count())
