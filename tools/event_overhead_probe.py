"""Cost of the per-conv HIP profiling events inside a timed loop (GPU box): ms per step of config 2 with / without
hcf_profile_convs, sync and lazy range-check policy (profiles/r03_notes.md: 3.3 ms before the events of back-to-back convs
were chained, 1.2 ms after)."""
import sys, time, torch, contextlib
sys.path.insert(0, '.')
from hcflow_amd import HCFlowNet_SR, preset, make_params
cfg = preset("SR_DF2K_4X")
with contextlib.redirect_stdout(sys.stderr):
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
net.load_state_dict(make_params(cfg, 1234), strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__: m.inited = True
net = net.cuda().eval()
lr = torch.rand(16, 3, 160, 160).cuda()
eng = net.engine()
def run(n, prof):
    eng.profile_convs(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(n): net(lr=lr, eps_std=0.8, reverse=True, seed=i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    eng.profile_convs(False); eng.conv_time(0, 0, reset=True)
    return dt * 1e3
run(3, False)
for rep in range(2):
    print("events off %.2f ms/step   events on %.2f ms/step" % (run(8, False), run(8, True)))
net.set_range_check("lazy")
print("lazy, events off %.2f  on %.2f" % (run(8, False), run(8, True)))
