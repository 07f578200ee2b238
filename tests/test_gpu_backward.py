"""-m gpu: backward kernels of the training path against torch.autograd on the CPU
(fp64 evaluation of the same op)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_ops import _gen

pytestmark = pytest.mark.gpu

# B, H, W, source channels, source upsample log2, cout, k
BWD_CASES = [
    (2, 16, 32, [32], [0], 32, 3),
    (1, 24, 40, [64, 96], [0, 0], 32, 3),          # RDB conv4 shape, ragged tile edges
    (2, 16, 16, [64, 128], [0, 0], 64, 3),         # RDB conv5
    (1, 16, 32, [6, 64], [0, 0], 64, 3),           # FCN conv1: 6 + cond channels
    (1, 16, 32, [3, 128], [0, 1], 64, 3),          # conv_first: second source read through an upsample
    (2, 12, 20, [64], [0], 64, 1),                 # FCN conv2 (1x1)
    (1, 9, 33, [64], [0], 12, 3),                  # Conv2dZeros head, 12 outputs, odd sizes
    (1, 16, 32, [21, 10], [0, 0], 22, 3),          # channel tails
]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))


@pytest.mark.parametrize("case", BWD_CASES)
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_conv2d_backward(case, precision):
    from hcflow_amd import ops
    B, H, W, cs, ups, cout, k = case
    g = _gen(sum(cs) + cout + H)
    srcs = [torch.randn(B, c, H >> u, W >> u, generator=g) for c, u in zip(cs, ups)]
    cin = sum(cs)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    # reference: fp64 autograd
    s64 = [s.double().requires_grad_(True) for s in srcs]
    w64 = w.double().requires_grad_(True)
    b64 = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(s64, ups)], 1)
    y = F.conv2d(x, w64, b64, 1, k // 2)
    y.backward(gy.double())
    ops.set_precision(precision)
    try:
        dsrcs, dw, db = ops.conv2d_backward([s.cuda() for s in srcs], w, gy.cuda(), ups)
    finally:
        ops.set_precision("exact")
    for d, s in zip(dsrcs, s64):
        assert _rel(d.cpu(), s.grad) <= 3e-6, (case, precision, _rel(d.cpu(), s.grad))
    assert _rel(dw, w64.grad) <= 3e-6, (case, _rel(dw, w64.grad))
    assert _rel(db, b64.grad) <= 1e-6


GRADS = ["grad_sr4_tiny", "grad_sr8_tiny"]


@pytest.mark.parametrize("name", GRADS)
def test_nll_step_gradients_match_reference(name):
    """One NLL step of HCFlow_SR_model.optimize_parameters (:195-199) through the drop-in module: nll and
    d nll / d parameter for every tensor of the net against the reference-generated fixture."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import param_spec
    from tests.util import load_golden, params_for, t
    from tests.test_oracle_golden import check_grads_against_fixture
    g = load_golden(name)
    cfg, p = params_for(g)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train()
    lr_hat, nll = net(hr=t(g["hr"]).cuda(), lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
    assert abs(float(nll.detach()) - float(g["fwd_nll"])) <= 2e-4 * max(1.0, abs(float(g["fwd_nll"])) / 100)
    assert float((lr_hat.cpu() - t(g["fwd_lr"])).abs().max()) <= 1e-4
    (nll * 1.0).backward()
    sd = dict(net.named_parameters())
    grads = [np.zeros(tuple(sd[k].shape), np.float32) if sd[k].grad is None else sd[k].grad.cpu().numpy()
             for k, _, _ in param_spec(cfg)]
    assert all(np.isfinite(x).all() for x in grads)
    check_grads_against_fixture(g, grads)
