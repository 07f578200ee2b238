"""Debug aid: hcf_aux_conv2d / hcf_aux_conv2d_backward on single layers against fp64 torch (y, dx, dw, db) and run-to-run bit
identity (GPU box); the test form is tests/test_gpu_gan.py::test_aux_conv_op_matches_torch."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from hcflow_amd import gan
torch.manual_seed(0)
work = {}
for (cin, cout, H, W, act) in [(64, 64, 64, 96, 0), (64, 64, 64, 96, 1), (128, 128, 32, 48, 0), (128, 128, 32, 48, 1), (3, 64, 64, 96, 1), (512, 512, 8, 12, 1), (64, 128, 32, 48, 1)]:
    x = torch.randn(2, cin, H, W, device="cuda") ; w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda") * 0.05
    res = []
    for rep in range(2):
        xn = gan._nhwc(x).requires_grad_(True); wn = w.clone().requires_grad_(True); bn = b.clone().requires_grad_(True)
        y = gan._ConvNHWC.apply(xn, wn, bn, act, 0, work, [])
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda()
        (y * gy).sum().backward()
        res.append((y.detach().clone(), xn.grad.clone(), wn.grad.clone(), bn.grad.clone()))
    xd = x.double().requires_grad_(True); wd = w.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd, 1, 1)
    if act == 1: yd = F.relu(yd)
    gyd = gy[..., :cout].permute(0, 3, 1, 2).double()
    (yd * gyd).sum().backward()
    y0, gx0, gw0, gb0 = res[0]
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    print("cin %3d cout %3d %dx%d act %d: y %.1e  dx %.1e  dw %.1e  db %.1e | run-to-run dx equal %s dw equal %s" % (
        cin, cout, H, W, act, rel(y0[..., :cout].permute(0, 3, 1, 2), yd), rel(gx0[..., :cin].permute(0, 3, 1, 2), xd.grad), rel(gw0, wd.grad), rel(gb0, bd.grad),
        bool(torch.equal(res[0][1], res[1][1])), bool(torch.equal(res[0][2], res[1][2]))))
