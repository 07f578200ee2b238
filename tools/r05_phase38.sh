#!/bin/bash
# element-major gauss_sample kernel: parity (bit-identical draws), same-box A/B against the previous build
O=gpurun_out/r05_p38
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_engine.py tests/test_gpu_callers.py tests/test_gpu_real.py -m gpu -q -x 2>&1 | grep -v "^shapes" | grep -E "passed|failed|FAILED|rror" | tail -4 | tee $O/pytest.log
python - <<'PY' 2>&1 | grep -v "^shapes" | tail -3
# bit-identity of the device draws between the two builds (same seed): full-size config-2 shapes
import os, subprocess, sys, torch
code = r'''
import sys, torch
sys.path.insert(0, ".")
from hcflow_amd import HCFlowNet_SR, preset, make_params
cfg = preset("SR_4X_tiny"); net = HCFlowNet_SR(opt=cfg.to_opt(), step=0); net.load_state_dict(make_params(cfg, 3), strict=True)
for m in net.modules():
    if "ActNorm" in type(m).__name__: m.inited = True
net = net.cuda().eval()
lr = torch.rand(5, 3, 24, 40, generator=torch.Generator().manual_seed(1)).cuda()
with torch.no_grad():
    out = net(lr=lr, eps_std=0.8, reverse=True, seed=77)
torch.save(out.cpu(), sys.argv[1])
'''
open("/tmp/draw.py", "w").write(code)
env = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/draw.py", "/tmp/new.pt"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
env["HCFLOW_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "build/ab/libhcflow_hip_base.so")
subprocess.check_call([sys.executable, "/tmp/draw.py", "/tmp/base.pt"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
a, b = torch.load("/tmp/new.pt"), torch.load("/tmp/base.pt")
print("seeded samples of the two builds bit-identical:", bool(torch.equal(a, b)), float((a - b).abs().max()))
PY
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export HCFLOW_LIB=$GRAFT_REPO_ROOT/build/ab/libhcflow_hip_base.so; else unset HCFLOW_LIB; fi
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-other-precision --no-exact-check --no-other-configs > $O/$lib.json 2> $O/$lib.err
  python - <<PY
import json
j=json.loads(open("$O/$lib.json").read().strip().splitlines()[-1])
print("$lib:", j["value"], j["ms_per_step"], "single", j["single_stream"]["value"], j["single_stream"]["ms_per_step"], "convs", j["roofline"]["all_convs"]["ms_per_step"])
PY
done
done
