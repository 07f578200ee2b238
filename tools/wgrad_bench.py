"""Time the weight-gradient kernel on one conv shape through the op API under rocprofv3 (kernel time only):
    rocprofv3 --kernel-trace --stats -d out -- python tools/wgrad_bench.py B H W cin cout [f16x3|exact]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hcflow_amd import ops  # noqa: E402

B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
ops.set_precision(sys.argv[6] if len(sys.argv) > 6 else "f16x3")      # f16x3: matrix-core kernel; exact: fp32 MFMA kernel
g = torch.Generator().manual_seed(1)
x = torch.randn(B, cin, H, W, generator=g).cuda()
w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
gy = torch.randn(B, cout, H, W, generator=g).cuda()
for _ in range(3):
    ops.conv2d_backward([x], w, gy, need_input_grads=False)
torch.cuda.synchronize()
