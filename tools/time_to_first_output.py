import sys, time, contextlib, torch
sys.path.insert(0, "/root/repo")
t0 = time.perf_counter()
from hcflow_amd import HCFlowNet_SR, preset, make_params
cfg = preset("SR_DF2K_4X")
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
t1 = time.perf_counter()
net.load_state_dict(make_params(cfg, 1), strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__: m.inited = True
t2 = time.perf_counter()
net = net.cuda().eval()
lr = torch.rand(1, 3, 160, 160).cuda()
torch.cuda.synchronize(); t3 = time.perf_counter()
with torch.no_grad():
    out = net(lr=lr, eps_std=0.8, reverse=True)
torch.cuda.synchronize(); t4 = time.perf_counter()
with torch.no_grad():
    out = net(lr=lr, eps_std=0.8, reverse=True)
torch.cuda.synchronize(); t5 = time.perf_counter()
print("import+construct %.2f s, make/load params %.2f s, .cuda() %.2f s, FIRST call (engine build: pack + upload) %.2f s, second call %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
