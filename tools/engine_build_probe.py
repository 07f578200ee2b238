import sys, time, contextlib, torch
sys.path.insert(0, ".")
from hcflow_amd import HCFlowNet_SR, preset, make_params, _lib
cfg = preset("SR_DF2K_4X")
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(make_params(cfg, 1), strict=True)
net = net.cuda().eval()
torch.cuda.synchronize()
named = net._tensors()
t0 = time.perf_counter()
cpu = [(k, t.detach().to("cpu", torch.float32).contiguous()) for k, t in named]
t1 = time.perf_counter()
eng = _lib.Engine(cfg)
for k, t in cpu:
    eng.set_param(k, t)
t2 = time.perf_counter()
eng.finalize(0)
torch.cuda.synchronize()
t3 = time.perf_counter()
for k, p in named:
    eng.bind_param_device(k, p.data_ptr())
t4 = time.perf_counter()
print("D2H copies %.3f s, set_param %.3f s, finalize %.3f s, bind %.3f s" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3))
