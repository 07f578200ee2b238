"""Run the f16x3 inverse pass several times: are the outputs bit-identical, and how far from the exact kernels?
    python tools/dbg_det.py PRESET B LRSIZE [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from hcflow_amd import HCFlowNet_SR, preset, make_params, eps_shapes  # noqa: E402

name, B, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cfg = preset(name)
p = make_params(cfg, 1234)
net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(p, strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__:
        m.inited = True
net = net.cuda().eval()
g = torch.Generator().manual_seed(5)
lr = torch.rand(B, 3, h, h, generator=g).cuda()
eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, h, h)]
with torch.no_grad():
    ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
    net.set_precision("f16x3")
    outs = [net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False) for _ in range(reps)]
same = [bool(torch.equal(o, outs[0])) for o in outs]
print(name, B, h, {k: v for k, v in os.environ.items() if k.startswith("HCF_")}, "bit-identical:", same,
      "max diff vs exact %.2e" % max(float((o - ex).abs().max()) for o in outs))
