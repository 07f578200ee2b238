"""-m gpu: BASELINE.json config 5 at its real shape -- one NLL training step (HCFlow_SR_model.optimize_parameters,
HCFlow_SR_model.py:184-205) of the FULL-DEPTH General-SR x4 net (K = 26, 13 + 13 steps per level, RRDB trunks 7 + 7), per-GPU
batch 16 of 160x160 HR patches, in both conv precisions; plus oracle-autograd parity of the full-depth net on a small patch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PRESET, SEED = "SR_DF2K_4X", 1234


def _net(precision):
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params
    cfg = preset(PRESET)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params(PRESET, SEED), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    return cfg, net.to("cuda:0").train().set_precision(precision)


def _batch(B, hr_size, seed):
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(B, 3, hr_size, hr_size, generator=g) * 0.8 + 0.1
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g)
    return hr, lr, noise


def _step(net, hr, lr, noise):
    net.zero_grad(set_to_none=True)
    _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
    nll.backward()
    torch.cuda.synchronize()
    return float(nll.detach()), [p.grad.detach().clone() for p in net.parameters()]


@pytest.fixture(scope="module")
def full_step_results():
    """config-5 shape, both precisions, two runs each (shared by the tests below: ~4 full-depth steps in total)."""
    hr, lr, noise = _batch(16, 160, 21)
    hr, lr, noise = hr.cuda(), lr.cuda(), noise.cuda()
    out = {}
    for prec in ("exact", "f16x3"):
        _, net = _net(prec)
        first = _step(net, hr, lr, noise)
        second = _step(net, hr, lr, noise)
        out[prec] = (first, second, net.engine().fallback_count())
        del net
        torch.cuda.empty_cache()
    return out


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_config5_step_is_finite_and_bit_reproducible(full_step_results, precision):
    (nll1, g1), (nll2, g2), fallbacks = full_step_results[precision]
    assert np.isfinite(nll1) and all(bool(torch.isfinite(x).all()) for x in g1)
    assert any(float(x.abs().max()) > 0 for x in g1)
    assert fallbacks == 0
    # fixed-order reductions everywhere (log-det partials, weight-gradient split-K, per-channel parameter sums)
    assert nll1 == nll2
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


def test_config5_step_f16x3_matches_exact(full_step_results):
    from hcflow_amd.config import param_spec, preset
    (nll_e, ge), _, _ = full_step_results["exact"]
    (nll_f, gf), _, _ = full_step_results["f16x3"]
    assert abs(nll_e - nll_f) <= 1e-4 * max(1.0, abs(nll_e))
    gmax = max(float(x.norm()) for x in ge)
    keys = [k for k, _, _ in param_spec(preset(PRESET))]
    worst = (0.0, None)
    for k, a, b in zip(keys, ge, gf):
        err = float((a - b).norm()) / max(float(a.norm()), 1e-6 * gmax)
        if err > worst[0]:
            worst = (err, k)
    assert worst[0] <= 2e-4, worst


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_config5_optimizer_step_updates_the_engine(precision):
    """nll.backward() -> clip_grad_norm_ -> Adam.step() -> the next forward sees the new parameters (device-side refresh of
    every pack of the full-depth net) and the loss falls on the same batch."""
    _, net = _net(precision)
    hr, lr, noise = (t.cuda() for t in _batch(16, 160, 22))
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-5, betas=(0.9, 0.99))
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        total = torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)
        assert bool(torch.isfinite(total))
        opt.step()
        losses.append(float(nll.detach()))
    assert all(np.isfinite(x) for x in losses)
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_full_depth_gradients_match_oracle_autograd(precision):
    """Full-depth net, B = 2, HR 32x32: nll and d nll / d parameter for all 1478 tensors against torch.autograd through the
    CPU oracle (oracle/hcflow_oracle.py is pinned to the reference's gradients on the tiny nets, tests/test_oracle_golden.py)."""
    from oracle import hcflow_oracle as O
    from hcflow_amd.config import param_spec
    from tests.util import cached_params
    cfg, net = _net(precision)
    hr, lr, noise = _batch(2, 32, 23)
    q = {k: v.clone().requires_grad_(True) for k, v in cached_params(PRESET, SEED).items()}
    _, nll_o = O.sr_forward(hr, lr, q, cfg, noise=noise)
    nll_o.backward()
    nll, grads = _step(net, hr.cuda(), lr.cuda(), noise.cuda())
    assert abs(nll - float(nll_o.detach())) <= 2e-4 * max(1.0, abs(float(nll_o.detach())) / 100)
    keys = [k for k, _, _ in param_spec(cfg)]
    gmax = max(float(q[k].grad.norm()) for k in keys if q[k].grad is not None)
    worst = (0.0, None)
    for k, g in zip(keys, grads):
        ref = q[k].grad
        if ref is None:
            continue
        err = float((g.cpu() - ref).norm()) / max(float(ref.norm()), 1e-6 * gmax)
        if err > worst[0]:
            worst = (err, k)
    assert worst[0] <= 3e-4, worst
