import sys, contextlib, torch
sys.path.insert(0, "/root/repo")
from hcflow_amd import HCFlowNet_SR, preset, make_params
cfg = preset("SR_DF2K_4X")
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(make_params(cfg, 1234), strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__: m.inited = True
net = net.cuda().train().set_precision("f16x3")
hr = torch.rand(16, 3, 160, 160).cuda()
lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2.5e-4, betas=(0.9, 0.99))
def step():
    opt.zero_grad(set_to_none=True)
    _, nll = net(hr=hr, lr=lr, reverse=False)
    nll.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)[:25]
for e in rows: print("%-45s count %6d  cpu_total %.2f ms" % (e.key[:45], e.count, e.cpu_time_total / 1e3))
