"""Single conv, f16x3 vs exact kernels, on a large grid; prints where (if anywhere) they disagree and whether the
f16x3 result is reproducible.  python tools/dbg_conv_cmp.py B H W cin cout [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from hcflow_amd import ops  # noqa: E402

B, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
g = torch.Generator().manual_seed(3)
x = torch.randn(B, cin, H, W, generator=g).cuda()
w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
ops.set_precision("exact")
ex = ops.conv2d([x], w)
if os.environ.get("HCF_ABLATE"):
    from hcflow_amd import _lib
    assert _lib.load().hcf_debug_set_ablation(int(os.environ["HCF_ABLATE"])) == 0
ops.set_precision("f16x3")
outs = [ops.conv2d([x], w) for _ in range(reps)]
ops.set_precision("exact")
for i, o in enumerate(outs):
    d = (o - ex).abs()
    bad = (d > 1e-3).nonzero()
    print("rep %d: max diff %.3e  bad elements %d  identical to rep0: %s" % (i, float(d.max()), bad.shape[0], bool(torch.equal(o, outs[0]))))
    if bad.shape[0]:
        b, c, y, xx = bad.T
        print("   batch idx", sorted(set(b.tolist()))[:16], " tile rows", sorted(set((y // 8).tolist()))[:12], "... tile cols", sorted(set((xx // 32).tolist()))[:12])
        print("   channels", sorted(set(c.tolist()))[:40])
        print("   y%8", sorted(set((y % 8).tolist())), " x%32", sorted(set((xx % 32).tolist())))
        tiles = set(zip(b.tolist(), (y // 8).tolist(), (xx // 32).tolist()))
        print("   distinct bad tiles", len(tiles), "of", B * ((H + 7) // 8) * ((W + 31) // 32), " first", sorted(tiles)[:8])
