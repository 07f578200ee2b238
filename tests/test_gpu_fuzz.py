"""-m gpu: seeded fuzzing of the hot-path entry points against fp64 PyTorch evaluations / the oracle: random conv
configurations (1-3 source windows, nearest-upsampled sources, residual epilogues, activations, ragged sizes, channel
tails), random conv-backward shapes, and random network configurations (steps per level, steps after the split, RRDB
counts)."""
import dataclasses

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hcflow_oracle as O
from hcflow_amd.config import preset, eps_shapes
from hcflow_amd.params import make_params

pytestmark = pytest.mark.gpu

ACTS = [None, "relu", "lrelu"]


def _rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))


def _conv_case(rng):
    n_src = int(rng.integers(1, 4))
    ups = [0] + [int(rng.integers(1, 3)) if rng.random() < 0.3 else 0 for _ in range(n_src - 1)]
    m = 1 << max(ups)
    H = max(1, int(rng.integers(1, 12))) * m
    W = max(1, int(rng.integers(1, 24))) * m
    cs = [int(rng.choice([1, 3, 6, 10, 21, 32, 64, 96, 128])) for _ in range(n_src)]
    cout = int(rng.choice([3, 6, 12, 22, 32, 48, 64, 96]))
    k = 3 if rng.random() < 0.85 else 1
    return dict(B=int(rng.integers(1, 4)), H=H, W=W, cs=cs, ups=ups, cout=cout, k=k, act=ACTS[int(rng.integers(0, 3))],
                res=int(rng.integers(0, 3)))


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_conv2d_fuzz(seed, precision):
    from hcflow_amd import ops
    rng = np.random.default_rng(1000 + seed)
    c = _conv_case(rng)
    g = torch.Generator().manual_seed(seed)
    B, H, W, k, cout = c["B"], c["H"], c["W"], c["k"], c["cout"]
    srcs = [torch.randn(B, n, H >> u, W >> u, generator=g) for n, u in zip(c["cs"], c["ups"])]
    cin = sum(c["cs"])
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.2
    scale = torch.exp(torch.randn(cout, generator=g) * 0.2)
    act = c["act"] if c["res"] == 0 else None            # the nets never combine an activation with a residual
    r1 = torch.randn(B, cout, H, W, generator=g) if c["res"] >= 1 else None
    r2 = torch.randn(B, cout, H, W, generator=g) if c["res"] >= 2 else None
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(srcs, c["ups"])], 1)
    ref = (F.conv2d(x.double(), w.double(), None, 1, k // 2) + bias.double().view(1, -1, 1, 1)) * scale.double().view(1, -1, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    if r1 is not None:
        ref = ref * 0.2 + r1.double()
    if r2 is not None:
        ref = ref * 0.3 + r2.double()
    ops.set_precision(precision)
    try:
        out = ops.conv2d([s.cuda() for s in srcs], w, bias, scale, act, c["ups"], res1=None if r1 is None else r1.cuda(),
                         rs1=0.2, res2=None if r2 is None else r2.cuda(), rs2=0.3)
    finally:
        ops.set_precision("exact")
    assert _rel(out, ref) <= 4e-6, (c, _rel(out, ref))


@pytest.mark.parametrize("seed", range(12))
def test_conv2d_backward_fuzz(seed):
    from hcflow_amd import ops
    rng = np.random.default_rng(2000 + seed)
    c = _conv_case(rng)
    c["cout"] = min(c["cout"], 64)
    g = torch.Generator().manual_seed(100 + seed)
    B, H, W, k, cout = c["B"], c["H"], c["W"], c["k"], c["cout"]
    srcs = [torch.randn(B, n, H >> u, W >> u, generator=g) for n, u in zip(c["cs"], c["ups"])]
    cin = sum(c["cs"])
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    s64 = [s.double().requires_grad_(True) for s in srcs]
    w64 = w.double().requires_grad_(True)
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(s64, c["ups"])], 1)
    F.conv2d(x, w64, None, 1, k // 2).backward(gy.double())
    dsrcs, dw, db = ops.conv2d_backward([s.cuda() for s in srcs], w, gy.cuda(), c["ups"])
    for d, s in zip(dsrcs, s64):
        assert _rel(d, s.grad) <= 4e-6, (c, _rel(d, s.grad))
    assert _rel(dw, w64.grad) <= 4e-6, (c, _rel(dw, w64.grad))
    assert _rel(db, gy.double().sum(dim=(0, 2, 3))) <= 2e-6


@pytest.mark.parametrize("seed", range(6))
def test_random_network_configurations_match_oracle(seed):
    """Depth / split variations of the SR nets (K steps per level, how many of them act on the split half, RRDB
    counts): the engine builds its layer plan from the same options as the reference constructor; inverse and NLL
    forward against the oracle on the same seeded weights."""
    from hcflow_amd import HCFlowNet_SR
    rng = np.random.default_rng(3000 + seed)
    base = preset("SR_4X_tiny" if seed % 2 == 0 else "SR_8X_tiny")
    L = base.L
    K = [int(rng.integers(1, 5)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 3)), int(rng.integers(1, 3))))
    cfg.validate()
    p = make_params(cfg, 500 + seed)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(seed)
    h, w = int(rng.integers(2, 6)) * 2, int(rng.integers(2, 7)) * 2
    lr = torch.rand(2, 3, h, w, generator=g)
    eps = [torch.randn(s, generator=g) * 0.7 for s in eps_shapes(cfg, 2, h, w)]
    hr = torch.rand(2, 3, h * cfg.scale, w * cfg.scale, generator=g)
    noise = torch.rand(hr.shape, generator=g)
    want = O.sr_inverse(lr, p, cfg, 0.7, eps=eps, clamp=False)
    lr_o, nll_o = O.sr_forward(hr, lr, p, cfg, noise=noise)
    with torch.no_grad():
        got = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.7, eps=[e.cuda() for e in eps], clamp=False)
        lr_g, nll_g = net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())
    sc = max(1.0, float(want.abs().max()))
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * sc, (K, after, float((got.cpu() - want).abs().max()))
    assert float((lr_g.cpu() - lr_o).abs().max()) <= 1e-4
    assert abs(float(nll_g) - float(nll_o)) <= 2e-4 * max(1.0, abs(float(nll_o)) / 100)


def _grad_cmp(net, q, cfg, rtol):
    from hcflow_amd.config import param_spec
    sd = dict(net.named_parameters())
    refs = {k: q[k].grad for k, _, _ in param_spec(cfg) if torch.is_tensor(q[k]) and q[k].grad is not None}
    gmax = max(float(v.norm()) for v in refs.values())
    worst = 0.0
    for k, ref in refs.items():
        have = sd[k].grad.cpu() if sd[k].grad is not None else torch.zeros_like(ref)
        worst = max(worst, float((have - ref).norm()) / max(float(ref.norm()), 2e-5 * gmax))
    assert worst <= rtol, worst


@pytest.mark.parametrize("seed", range(4))
def test_random_network_gradients_match_oracle_autograd(seed):
    """NLL-step and reverse-path gradients of randomly configured SR nets (steps per level, steps after the split,
    RRDB counts, ragged sizes) against autograd through the oracle."""
    from hcflow_amd import HCFlowNet_SR
    rng = np.random.default_rng(4000 + seed)
    base = preset("SR_4X_tiny" if seed % 2 == 0 else "SR_8X_tiny")
    K = [int(rng.integers(1, 4)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 2)), int(rng.integers(1, 3))))
    cfg.validate()
    p = make_params(cfg, 700 + seed)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train()
    g = torch.Generator().manual_seed(seed)
    h, w = int(rng.integers(2, 5)) * 2, int(rng.integers(2, 6)) * 2
    lr = torch.rand(2, 3, h, w, generator=g)
    hr = torch.rand(2, 3, h * cfg.scale, w * cfg.scale, generator=g) * 0.6 + 0.2
    noise = torch.rand(hr.shape, generator=g)
    eps = [torch.randn(s, generator=g) * 0.5 for s in eps_shapes(cfg, 2, h, w)]
    # NLL step
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    O.sr_forward(hr, lr, q, cfg, noise=noise)[1].backward()
    net.zero_grad()
    net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())[1].backward()
    _grad_cmp(net, q, cfg, 5e-4)
    # reverse path with an L1 pixel loss
    q = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    F.l1_loss(O.sr_inverse(lr, q, cfg, 0.5, eps=eps), hr).backward()
    net.zero_grad()
    F.l1_loss(net(lr=lr.cuda(), eps_std=0.5, reverse=True, eps=[e.cuda() for e in eps]), hr.cuda()).backward()
    _grad_cmp(net, q, cfg, 5e-3)          # L1's sign() and the clamp mask make single-pixel flips visible


@pytest.mark.parametrize("seed", range(3))
def test_random_rescaling_configurations_match_oracle(seed):
    """Rescaling net variations (steps per level / after the split, RRDB counts): forward and inverse vs the oracle."""
    from hcflow_amd import HCFlowNet_Rescaling
    rng = np.random.default_rng(5000 + seed)
    base = preset("Rescaling_4X_tiny")
    K = [int(rng.integers(1, 5)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 3)), int(rng.integers(1, 3))))
    cfg.validate()
    p = make_params(cfg, 900 + seed)
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval()
    g = torch.Generator().manual_seed(seed)
    h, w = int(rng.integers(2, 6)) * 2, int(rng.integers(2, 7)) * 2
    hr = torch.rand(2, 3, h * 4, w * 4, generator=g)
    lr = torch.rand(2, 3, h, w, generator=g)
    eps = [torch.randn(s, generator=g) for s in eps_shapes(cfg, 2, h, w)]
    lo, z1o, z2o = O.rescale_forward(hr, p, cfg)
    inv_o = O.rescale_inverse(lr, p, cfg, 1.0, eps=eps, clamp=False)
    with torch.no_grad():
        lg, z1g, z2g = net(hr=hr.cuda(), reverse=False)
        inv_g = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=1.0, eps=[e.cuda() for e in eps], clamp=False)
    assert float((lg.cpu() - lo).abs().max()) <= 1e-4
    assert float((z1g.cpu() - z1o).abs().max()) <= 1e-4 * max(1.0, float(z1o.abs().max()))
    assert float((z2g.cpu() - z2o).abs().max()) <= 1e-4 * max(1.0, float(z2o.abs().max()))
    assert float((inv_g.cpu() - inv_o).abs().max()) <= 1e-4 * max(1.0, float(inv_o.abs().max()))


# ---- the same random configurations on the DEFAULT precision (f16x3), at sizes of several tiles: the schedules that only exist
# there (fat dense-block launches, the Winograd forms of the FCN / DenseBlock coupling nets, fused tails) against the oracle.
# HCF_FUZZ_SEEDS=n runs a longer campaign by hand.
import os  # noqa: E402

_NF = int(os.environ.get("HCF_FUZZ_SEEDS", "6"))
_SZ = int(os.environ.get("HCF_FUZZ_SIZE", "1"))            # multiplies the LR sizes (by hand: grids of several rounds)


@pytest.mark.parametrize("seed", range(_NF))
def test_random_sr_configurations_f16x3_match_oracle(seed):
    from hcflow_amd import HCFlowNet_SR
    rng = np.random.default_rng(6000 + seed)
    base = preset("SR_4X_tiny" if seed % 3 else "SR_8X_tiny")
    K = [int(rng.integers(1, 5)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 3)), int(rng.integers(1, 3))))
    cfg.validate()
    p = make_params(cfg, 1500 + seed)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval().set_precision("f16x3")
    g = torch.Generator().manual_seed(seed)
    B = int(rng.integers(1, 4))
    h, w = int(rng.integers(5, 24)) * 2 * _SZ, int(rng.integers(5, 30)) * 2 * _SZ
    lr = torch.rand(B, 3, h, w, generator=g)
    eps = [torch.randn(s, generator=g) * 0.7 for s in eps_shapes(cfg, B, h, w)]
    hr = torch.rand(B, 3, (h // 2) * cfg.scale, (w // 2) * cfg.scale, generator=g)          # NLL pass at half the linear size
    lrh = torch.rand(B, 3, h // 2, w // 2, generator=g)
    noise = torch.rand(hr.shape, generator=g)
    want = O.sr_inverse(lr, p, cfg, 0.7, eps=eps, clamp=False)
    lr_o, nll_o = O.sr_forward(hr, lrh, p, cfg, noise=noise)
    with torch.no_grad():
        got = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.7, eps=[e.cuda() for e in eps], clamp=False)
        lr_g, nll_g = net(hr=hr.cuda(), lr=lrh.cuda(), reverse=False, noise=noise.cuda())
    assert net.engine().fallback_count() == 0
    sc = max(1.0, float(want.abs().max()))
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * sc, (K, after, B, h, w, float((got.cpu() - want).abs().max()))
    assert float(((lr_g.cpu() - lr_o).abs() > 0.5 / 255).float().mean()) <= 1e-3      # quantised LR^: a level may flip on a boundary
    assert abs(float(nll_g) - float(nll_o)) <= 2e-4 * max(1.0, abs(float(nll_o)) / 100)


@pytest.mark.parametrize("seed", range(max(2, _NF // 2)))
def test_random_rescaling_configurations_f16x3_match_oracle(seed):
    from hcflow_amd import HCFlowNet_Rescaling
    rng = np.random.default_rng(7000 + seed)
    base = preset("Rescaling_4X_tiny")
    K = [int(rng.integers(1, 5)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 3)), int(rng.integers(1, 3))))
    cfg.validate()
    p = make_params(cfg, 1900 + seed)
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval().set_precision("f16x3")
    g = torch.Generator().manual_seed(seed)
    B = int(rng.integers(1, 3))
    h, w = int(rng.integers(5, 20)) * 2 * _SZ, int(rng.integers(5, 26)) * 2 * _SZ
    hr = torch.rand(B, 3, h * 4, w * 4, generator=g)
    lr = torch.rand(B, 3, h, w, generator=g)
    eps = [torch.randn(s, generator=g) for s in eps_shapes(cfg, B, h, w)]
    lo, z1o, z2o = O.rescale_forward(hr, p, cfg)
    inv_o = O.rescale_inverse(lr, p, cfg, 1.0, eps=eps, clamp=False)
    with torch.no_grad():
        lg, z1g, z2g = net(hr=hr.cuda(), reverse=False)
        inv_g = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=1.0, eps=[e.cuda() for e in eps], clamp=False)
    assert net.engine().fallback_count() == 0
    assert float(((lg.cpu() - lo).abs() > 0.5 / 255).float().mean()) <= 1e-3
    assert float((z1g.cpu() - z1o).abs().max()) <= 1e-4 * max(1.0, float(z1o.abs().max()))
    assert float((z2g.cpu() - z2o).abs().max()) <= 1e-4 * max(1.0, float(z2o.abs().max()))
    assert float((inv_g.cpu() - inv_o).abs().max()) <= 1e-4 * max(1.0, float(inv_o.abs().max()))


@pytest.mark.parametrize("seed", range(4))
def test_random_lu_configurations_match_oracle_incl_gradients(seed):
    """LU-decomposed invertible convs (Permutations.py:41-57,78-92) in randomly configured SR nets: inverse, NLL and the NLL-step
    gradients (l, log_s, u among them) against the oracle / its autograd, module default precision."""
    from hcflow_amd import HCFlowNet_SR
    from tests.util import trainable
    rng = np.random.default_rng(8000 + seed)
    base = preset("SR_4X_tiny_LU" if seed % 2 == 0 else "SR_8X_tiny_LU")
    K = [int(rng.integers(1, 4)) for _ in range(len(base.K))]
    after = [int(rng.integers(0, K[l] + 1)) for l in range(len(base.after))]
    cfg = dataclasses.replace(base, K=K, after=after, rrdb_nb=(int(rng.integers(0, 2)), int(rng.integers(1, 3))))
    cfg.validate()
    assert cfg.lu
    p = make_params(cfg, 1100 + seed)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train()
    g = torch.Generator().manual_seed(seed)
    h, w = int(rng.integers(2, 5)) * 2, int(rng.integers(2, 6)) * 2
    lr = torch.rand(2, 3, h, w, generator=g)
    hr = torch.rand(2, 3, h * cfg.scale, w * cfg.scale, generator=g) * 0.6 + 0.2
    noise = torch.rand(hr.shape, generator=g)
    eps = [torch.randn(s, generator=g) * 0.6 for s in eps_shapes(cfg, 2, h, w)]
    with torch.no_grad():
        want = O.sr_inverse(lr, p, cfg, 0.6, eps=eps, clamp=False)
        got = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.6, eps=[e.cuda() for e in eps], clamp=False)
    assert float((got.cpu() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
    q = trainable(p, cfg)
    nll_o = O.sr_forward(hr, lr, q, cfg, noise=noise)[1]
    nll_o.backward()
    net.zero_grad()
    nll_g = net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())[1]
    assert abs(float(nll_g.detach()) - float(nll_o.detach())) <= 2e-4 * max(1.0, abs(float(nll_o.detach())) / 100)
    nll_g.backward()
    _grad_cmp(net, q, cfg, 5e-4)
