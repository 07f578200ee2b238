"""-m gpu: the fp32-equivalent f16x3 convolution path (hcf_conv_f16x3.hip) against the oracle and the
reference-generated fixtures, with the same tolerances as the exact path (1e-5 per conv, 1e-4 end
to end, NLL 1e-4 bits/dim), plus the range-overflow fallback."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hcflow_amd.config import preset, eps_shapes
from tests.util import load_golden, params_for, t, maxdiff, cached_params
from tests.test_gpu_ops import CONV_CASES, _rel, _gen
from tests.test_gpu_nets import build_net, _eps, NETS_SR, NETS_RS

pytestmark = pytest.mark.gpu


@pytest.fixture()
def f16x3_ops():
    from hcflow_amd import ops
    ops.set_precision("f16x3")
    yield ops
    ops.set_precision("exact")


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[5] <= 64])
def test_conv2d_f16x3(f16x3_ops, case):
    ops = f16x3_ops
    B, H, W, cs, ups, cout, k, act = case
    g = _gen(hash(case[:3]) % 1000 + cout)
    srcs = [torch.randn(B, c, H >> u, W >> u, generator=g) for c, u in zip(cs, ups)]
    cin = sum(cs)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    scale = torch.exp(torch.randn(cout, generator=g) * 0.1)
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(srcs, ups)], 1)
    ref = (F.conv2d(x.double(), w.double(), None, 1, k // 2) + bias.view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    out = ops.conv2d([s.cuda() for s in srcs], w, bias, scale, act, ups)
    # fp32-class accuracy against an fp64 evaluation
    assert _rel(out, ref.float()) <= 3e-6, (case, _rel(out, ref.float()))


def test_conv2d_f16x3_wide_dynamic_range(f16x3_ops):
    """Inputs spanning 1e-6 .. 1e3 and tiny weights: the scaled lo parts must not lose the small values."""
    ops = f16x3_ops
    g = _gen(77)
    x = torch.randn(1, 32, 16, 32, generator=g) * torch.logspace(-6, 3, 32).view(1, 32, 1, 1)
    w = torch.randn(32, 32, 3, 3, generator=g) * torch.logspace(-5, 0, 32).view(32, 1, 1, 1)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    out = ops.conv2d([x.cuda()], w)
    ex = F.conv2d(x, w, None, 1, 1)
    per_ch = (out.cpu().double() - ref).abs().amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3))
    per_ch32 = (ex.double() - ref).abs().amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3))
    assert float(per_ch.max()) <= 4e-6, (float(per_ch.max()), float(per_ch32.max()))


def test_conv2d_f16x3_out_of_range_is_reported(f16x3_ops):
    from hcflow_amd import _lib
    x = torch.ones(1, 16, 8, 32)
    x[0, 3, 2, 5] = 1.0e5                      # beyond the f16 range
    w = torch.randn(16, 16, 3, 3, generator=_gen(1)) * 0.1
    with pytest.raises(_lib.HcfError):
        f16x3_ops.conv2d([x.cuda()], w)


@pytest.mark.parametrize("name", NETS_SR + NETS_RS)
def test_inverse_f16x3_matches_reference(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = build_net(cfg, p).set_precision("f16x3")
    try:
        with torch.no_grad():
            for ti in (0, 1):
                tau = float(g["inv%d_tau" % ti])
                eps = _eps(g, "inv%d" % ti)
                raw = net.reverse_flow_diracLR(t(g["lr"]).cuda(), None, None, eps_std=tau, eps=eps, clamp=False)
                scale = max(1.0, float(np.abs(g["inv%d_raw" % ti]).max()))
                assert maxdiff(raw, g["inv%d_raw" % ti]) <= 1e-4 * scale, (name, ti, maxdiff(raw, g["inv%d_raw" % ti]))
        assert net.engine().fallback_count() == 0
    finally:
        net.set_precision("exact")


@pytest.mark.parametrize("name", NETS_SR)
def test_forward_nll_f16x3(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = build_net(cfg, p).set_precision("f16x3")
    try:
        with torch.no_grad():
            hr, noise = t(g["hr"]).cuda(), t(g["fwd_noise"]).cuda()
            _, nll_self = net(hr=hr, lr=t(g["fwd_lr"]).cuda(), reverse=False, noise=noise)
            assert abs(float(nll_self) - float(g["fwd_nll_self"])) <= 1e-4
    finally:
        net.set_precision("exact")


def test_f16x3_vs_exact_full_size_and_fallback():
    """160x160 LR (the bench shape, B=1): f16x3 vs the exact fp32 kernels, then an input that
    overflows f16 to exercise the automatic exact re-run."""
    cfg = preset("SR_4X_tiny")
    p = cached_params("SR_4X_tiny", 11)
    net = build_net(cfg, p).set_precision("exact")
    g = torch.Generator().manual_seed(8)
    lr = torch.rand(1, 3, 160, 160, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, 1, 160, 160)]
    with torch.no_grad():
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        net.set_precision("f16x3")
        try:
            fa = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert maxdiff(fa, ex) <= 2e-5 * max(1.0, float(ex.abs().max()))
            n0 = net.engine().fallback_count()
            big = lr.clone()
            big[0, 1, 7, 9] = 7.0e4                      # > 65504: not representable by the f16 hi part
            fb = net.reverse_flow_diracLR(big, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert net.engine().fallback_count() == n0 + 1
            net.set_precision("exact")
            eb = net.reverse_flow_diracLR(big, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert torch.allclose(fb, eb, rtol=0, atol=0, equal_nan=True)      # bit-identical re-run
        finally:
            net.set_precision("exact")


LARGE = [("SR_4X_tiny", 11, 4, 160), ("SR_8X_tiny", 12, 4, 80), ("Rescaling_4X_tiny", 13, 4, 160)]


@pytest.mark.parametrize("name,seed,B,size", LARGE)
def test_large_grid_f16x3_matches_exact_and_is_deterministic(name, seed, B, size):
    """Grids with several blocks per CU: the f16x3 path, with its fused epilogues, must agree with the exact
    kernels and be bit-identical from run to run (a timing-dependent fault in the fused flow-step tail once slipped
    through every small-size test: it only showed with >= 2 co-resident blocks)."""
    cfg = preset(name)
    p = cached_params(name, seed)
    net = build_net(cfg, p).set_precision("exact")
    g = torch.Generator().manual_seed(21)
    lr = torch.rand(B, 3, size, size, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, size, size)]
    with torch.no_grad():
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        net.set_precision("f16x3")
        try:
            runs = [net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False) for _ in range(3)]
        finally:
            net.set_precision("exact")
    assert maxdiff(runs[0], ex) <= 2e-5 * max(1.0, float(ex.abs().max()))
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    with torch.no_grad():
        ex2 = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
    assert torch.equal(ex, ex2)


def test_large_grid_forward_nll_f16x3_is_deterministic():
    """Same for the forward (encode + NLL) pass at B = 4, HR 320x320."""
    cfg = preset("SR_4X_tiny")
    p = cached_params("SR_4X_tiny", 11)
    net = build_net(cfg, p).set_precision("exact")
    g = torch.Generator().manual_seed(22)
    hr = torch.rand(4, 3, 320, 320, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bilinear", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    with torch.no_grad():
        z_e, nll_e = net(hr=hr, lr=lr, reverse=False, noise=noise)
        net.set_precision("f16x3")
        try:
            outs = [net(hr=hr, lr=lr, reverse=False, noise=noise) for _ in range(3)]
        finally:
            net.set_precision("exact")
    assert abs(float(outs[0][1]) - float(nll_e)) <= 1e-4
    for z, nll in outs[1:]:
        assert torch.equal(nll, outs[0][1])
        assert all(torch.equal(a, b) for a, b in zip(_as_list(z), _as_list(outs[0][0])))


def _as_list(z):
    return list(z) if isinstance(z, (list, tuple)) else [z]


def test_huge_image_takes_the_per_launch_fallbacks():
    """One 1024 x 1040 LR image (level 0: 2048 x 2080 pixels x 128-channel tensors = more than the 2 GB that the Winograd kernels'
    31-bit source offsets can address): the schedules that cannot fall back per launch (fat dense-block pairs, the FCN form with
    the 1x1 epilogue) must step aside beforehand, the plain Winograd launches fall back to the direct kernels one by one, and the
    pass still agrees with the exact kernels."""
    cfg = preset("SR_4X_tiny")
    net = build_net(cfg, cached_params("SR_4X_tiny", 11)).set_precision("exact")
    g = torch.Generator().manual_seed(23)
    h, w = 1024, 1040
    lr = torch.rand(1, 3, h, w, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, 1, h, w)]
    with torch.no_grad():
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        net.set_precision("f16x3")
        try:
            a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert net.engine().fallback_count() == 0
        finally:
            net.set_precision("exact")
    assert maxdiff(a, ex) <= 2e-5 * max(1.0, float(ex.abs().max()))
