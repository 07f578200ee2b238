"""-m gpu: end-to-end parity of the engine (through the drop-in classes and the C ABI) against the
reference-generated golden fixtures and the CPU oracle, plus size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import hcflow_oracle as O
from hcflow_amd.config import preset, eps_shapes
from tests.util import load_golden, params_for, t, maxdiff, cached_params

pytestmark = pytest.mark.gpu

# net_var_*: reference-generated fixtures for depth / split / trunk variants (the option space of tests/test_gpu_fuzz.py; since
# round 5 the oracle derives the layer structure from the state dict itself, not from the product's config.layer_plan)
NETS_SR = ["net_sr4_tiny", "net_sr8_tiny", "net_sr4_full", "net_sr8_full", "net_var_sr4_a", "net_var_sr4_b", "net_var_sr8_a",
           "net_var_sr8_b"]
NETS_RS = ["net_rescale_tiny", "net_rescale_full", "net_var_rescale_a", "net_var_rescale_b"]
_cache = {}


def build_net(cfg, p):
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    import os
    key = id(p)
    mode = os.environ.get("HCFLOW_PRECISION", "f16x3")      # the session-level parametrisation (conftest.py) or the shipped default
    if key in _cache:
        return _cache[key].set_precision(mode)              # a cached net may come from the other precision's run
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval()
    _cache.clear()
    _cache[key] = net
    return net


def _eps(g, pre):
    out, i = [], 0
    while "%s_eps%d" % (pre, i) in g.files:
        out.append(t(g["%s_eps%d" % (pre, i)]))
        i += 1
    return out


@pytest.mark.parametrize("name", NETS_SR + NETS_RS)
def test_inverse_matches_reference(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = build_net(cfg, p)
    with torch.no_grad():
        for ti in (0, 1):
            tau = float(g["inv%d_tau" % ti])
            eps = _eps(g, "inv%d" % ti)
            raw = net.reverse_flow_diracLR(t(g["lr"]).cuda(), None, None, eps_std=tau, eps=eps, clamp=False)
            scale = max(1.0, float(np.abs(g["inv%d_raw" % ti]).max()))
            assert maxdiff(raw, g["inv%d_raw" % ti]) <= 1e-4 * scale, (name, ti, maxdiff(raw, g["inv%d_raw" % ti]))
            out = net(lr=t(g["lr"]).cuda(), z=None, u=None, eps_std=tau, reverse=True, eps=eps)
            assert maxdiff(out, g["inv%d_out" % ti]) <= 1e-4
            assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0


@pytest.mark.parametrize("name", NETS_SR)
def test_sr_forward_nll_matches_reference(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = build_net(cfg, p)
    with torch.no_grad():
        hr, lr, noise = t(g["hr"]).cuda(), t(g["lr"]).cuda(), t(g["fwd_noise"]).cuda()
        lr_hat, nll, logdet, z = net.normal_flow_diracLR(hr, lr, noise=noise, return_internals=True)
        assert maxdiff(z, g["fwd_z"]) <= 1e-4
        d = (lr_hat.cpu() - t(g["fwd_lr"])).abs()
        assert float(d.max()) <= 1.0 / 255 + 1e-6 and float((d > 1e-6).float().mean()) < 0.01
        # self-consistent NLL (lr := LR^): bits/dim within 1e-4 of the reference (BASELINE.json)
        _, nll_self = net(hr=hr, lr=t(g["fwd_lr"]).cuda(), reverse=False, noise=noise)
        assert abs(float(nll_self) - float(g["fwd_nll_self"])) <= 1e-4, (float(nll_self), float(g["fwd_nll_self"]))
        assert abs(float(nll) - float(g["fwd_nll"])) <= 1e-5 * abs(float(g["fwd_nll"]))


@pytest.mark.parametrize("name", NETS_RS)
def test_rescale_forward_and_roundtrip(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = build_net(cfg, p)
    with torch.no_grad():
        lr_hat, z1, z2 = net(hr=t(g["hr"]).cuda(), reverse=False)
        assert maxdiff(lr_hat, g["fwd_lr"]) <= 1e-4
        assert maxdiff(z1, g["fwd_z1"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z1"]).max()))
        assert maxdiff(z2, g["fwd_z2"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z2"]).max()))
        rt = net(lr=t(g["rt_lrq"]).cuda(), eps_std=1.0, reverse=True, eps=_eps(g, "rt"))
        assert maxdiff(rt, g["rt_out"]) <= 1e-4


def test_oracle_parity_on_fresh_inputs():
    """HIP path vs oracle on new seeded inputs (not in the fixtures), ragged LR size, B=3."""
    cfg = preset("SR_4X_tiny")
    from tests.util import cached_params
    p = cached_params("SR_4X_tiny", 11)
    net = build_net(cfg, p)
    g = torch.Generator().manual_seed(99)
    lr = torch.rand(3, 3, 9, 35, generator=g)
    eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 3, 9, 35)]
    with torch.no_grad():
        ref = O.sr_inverse(lr, p, cfg, 0.8, eps, clamp=False)
        out = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.8, eps=eps, clamp=False)
    assert maxdiff(out, ref) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_rescale_encode_decode_roundtrip_property():
    """Size-independent property: forward (no clamp) -> inverse with eps := encoded latents
    reproduces the HR input (invertibility of the whole flow incl. the conditional prior)."""
    cfg = preset("Rescaling_4X_tiny")
    from tests.util import cached_params
    p = cached_params("Rescaling_4X_tiny", 13)
    net = build_net(cfg, p)
    g = torch.Generator().manual_seed(5)
    hr = torch.rand(2, 3, 96, 128, generator=g).cuda()
    with torch.no_grad():
        lr_raw, z1, z2 = net.normal_flow_diracLR(hr, clamp=False)
        back = net.reverse_flow_diracLR(lr_raw, None, None, eps_std=1.0, eps=[z2, z1], clamp=False)
    assert maxdiff(back, hr) <= 2e-4


def test_batch_independence_and_determinism():
    """Every op is per-sample: sample i of a batched call equals the B=1 call;
    tau=0 is deterministic; device-sampled eps follows torch.manual_seed."""
    cfg = preset("SR_4X_tiny")
    from tests.util import cached_params
    p = cached_params("SR_4X_tiny", 11)
    net = build_net(cfg, p)
    lr = torch.rand(4, 3, 16, 24, generator=torch.Generator().manual_seed(3)).cuda()
    with torch.no_grad():
        full = net(lr=lr, eps_std=0.0, reverse=True)
        for i in (0, 3):
            one = net(lr=lr[i:i + 1], eps_std=0.0, reverse=True)
            assert torch.equal(one[0], full[i])
        assert torch.equal(full, net(lr=lr, eps_std=0.0, reverse=True))
        torch.manual_seed(7)
        a = net(lr=lr, eps_std=0.8, reverse=True)
        torch.manual_seed(7)
        b = net(lr=lr, eps_std=0.8, reverse=True)
        c = net(lr=lr, eps_std=0.8, reverse=True)
        assert torch.equal(a, b) and not torch.equal(a, c)


def test_parameter_update_triggers_repack():
    cfg = preset("SR_4X_tiny")
    from tests.util import cached_params
    p = cached_params("SR_4X_tiny", 11)
    net = build_net(cfg, p)
    lr = torch.rand(1, 3, 8, 8, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        a = net(lr=lr, eps_std=0.0, reverse=True)
        net.flow.level0_condFlow.f.bias.add_(0.05)
        b = net(lr=lr, eps_std=0.0, reverse=True)
        net.flow.level0_condFlow.f.bias.sub_(0.05)
        c = net(lr=lr, eps_std=0.0, reverse=True)
    assert not torch.equal(a, b) and maxdiff(a, c) <= 1e-6


ANINIT = ["aninit_sr4_tiny", "aninit_sr8_tiny", "aninit_rescale_tiny"]


@pytest.mark.parametrize("name", ANINIT)
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_actnorm_data_init_pass(name, precision):
    """train() mode, every ActNorm zeroed and ``inited = False``: ONE forward pass fits bias / logs where the
    reference does (ActNorms.py:29-43,78-80), returns the reference's outputs, stores the fitted values in the
    module's parameters and flips ``inited``; the following eval() pass uses them."""
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    g = load_golden(name)
    cfg, p = params_for(g)
    keys = [str(k) for k in g["an_keys"]]
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    mods = dict(net.named_modules())
    with torch.no_grad():
        for k in keys:
            mods[k].bias.zero_()
            mods[k].logs.zero_()
            assert mods[k].inited is False
    net = net.to("cuda:0").train().set_precision(precision)
    hr = t(g["hr"]).cuda()
    with torch.no_grad():
        if cfg.sr:
            lr_hat, nll = net(hr=hr, lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
            assert abs(float(nll) - float(g["fwd_nll"])) <= 2e-4 * max(1.0, abs(float(g["fwd_nll"])) / 100)
        else:
            lr_hat, z1, z2 = net(hr=hr, reverse=False)
            assert maxdiff(z1, g["fwd_z1"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z1"]).max()))
            assert maxdiff(z2, g["fwd_z2"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z2"]).max()))
        assert maxdiff(lr_hat, g["fwd_lr"]) <= 1e-4
        for i, k in enumerate(keys):
            m = mods[k]
            assert m.inited is True
            assert maxdiff(m.bias.reshape(-1), g["an_bias_%d" % i]) <= 1e-5 * max(1.0, float(np.abs(g["an_bias_%d" % i]).max())), k
            assert maxdiff(m.logs.reshape(-1), g["an_logs_%d" % i]) <= 1e-5, k
        # second pass: nothing left to fit, same parameters -> same outputs (now possibly on the f16x3 kernels)
        net.eval()
        if cfg.sr:
            lr2, nll2 = net(hr=hr, lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
            assert abs(float(nll2) - float(nll)) <= 1e-4 * max(1.0, abs(float(nll)) / 100)
        else:
            lr2, _, _ = net(hr=hr, reverse=False)
        assert maxdiff(lr2, lr_hat) <= 1e-4
    net.set_precision("exact")


def test_actnorm_init_rules():
    """eval() never initialises (ActNorms.py:31-32); a non-zero bias only flips ``inited`` (:33-35), on either path; the
    reverse path refuses genuinely un-initialised (zero-bias) layers in train() mode instead of guessing."""
    from hcflow_amd import HCFlowNet_SR
    cfg = preset("SR_4X_tiny")
    p = cached_params("SR_4X_tiny", 11)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    net = net.to("cuda:0")
    g = torch.Generator().manual_seed(3)
    hr = torch.rand(1, 3, 32, 32, generator=g).cuda()
    lr = torch.rand(1, 3, 8, 8, generator=g).cuda()
    an = [m for m in net.modules() if "ActNorm" in type(m).__name__]
    before = [m.bias.detach().clone() for m in an]
    with torch.no_grad():
        net.eval()
        net(hr=hr, lr=lr, reverse=False)
        assert not any(m.inited for m in an)                       # eval: untouched, still un-initialised
        net.train()
        out = net(lr=lr, eps_std=0.5, reverse=True, seed=1)        # non-zero biases: the reference only flips the flags
        assert all(m.inited for m in an) and bool(torch.isfinite(out).all())
        for m in an:
            m.inited = False
        net(hr=hr, lr=lr, reverse=False)                           # same on the forward path: flags flip, values stay
    assert all(m.inited for m in an)
    assert all(torch.equal(m.bias, b) for m, b in zip(an, before))
    # genuinely un-initialised layers (zero bias) on the REVERSE path in train() mode: refused, not guessed
    with torch.no_grad():
        an[0].bias.zero_()
        net.invalidate()
        for m in an:
            m.inited = False
        with pytest.raises(NotImplementedError):
            net(lr=lr, eps_std=0.5, reverse=True)


# ---- BASELINE.json's full-size configurations, through size-independent properties ------------------------------
def _full_net(name, seed):
    cfg = preset(name)
    return cfg, build_net(cfg, cached_params(name, seed))


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_config2_full_size_batch_independence(precision):
    """Config 2 (SR x4, B = 16, LR 160^2 -> HR 640^2, tau = 0.8): every op of the path is per-sample, so sample k
    of the batched run must equal the same sample run alone (same injected eps) bit for bit."""
    cfg, net = _full_net("SR_DF2K_4X", 1234)
    g = torch.Generator().manual_seed(77)
    B, h = 16, 160
    lr = torch.rand(B, 3, h, h, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, h, h)]
    net.set_precision(precision)
    try:
        with torch.no_grad():
            full = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert bool(torch.isfinite(full).all())
            for k in (0, 9):
                one = net.reverse_flow_diracLR(lr[k:k + 1], None, None, eps_std=0.8, eps=[e[k:k + 1] for e in eps], clamp=False)
                assert torch.equal(one, full[k:k + 1]), k
    finally:
        net.set_precision("exact")


def test_config3_face_x8_tau_sweep():
    """Config 3 (Face x8, B = 32, LR 20^2 -> 160^2): tau = 0 is deterministic (z = mean) and independent of the
    injected noise; across the tau sweep the f16x3 kernels stay within 1e-4 of the exact ones."""
    cfg, net = _full_net("SR_CelebA_8X", 1234)
    g = torch.Generator().manual_seed(78)
    B, h = 32, 20
    lr = torch.rand(B, 3, h, h, generator=g).cuda()
    unit = [torch.randn(s, generator=g).cuda() for s in eps_shapes(cfg, B, h, h)]
    with torch.no_grad():
        a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.0, eps=[u * 0.0 for u in unit], clamp=False)
        b = net.reverse_flow_diracLR(lr, None, None, eps_std=0.0, eps=None, clamp=False)
        assert torch.equal(a, b)
        for tau in (0.2, 0.6, 1.0):
            eps = [u * tau for u in unit]
            net.set_precision("exact")
            ex = net.reverse_flow_diracLR(lr, None, None, eps_std=tau, eps=eps, clamp=False)
            net.set_precision("f16x3")
            try:
                fa = net.reverse_flow_diracLR(lr, None, None, eps_std=tau, eps=eps, clamp=False)
            finally:
                net.set_precision("exact")
            assert bool(torch.isfinite(ex).all())
            assert maxdiff(fa, ex) <= 1e-4 * max(1.0, float(ex.abs().max())), tau


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_config4_rescaling_shard_roundtrip(precision):
    """Config 4 (rescaling x4, 8 images of 640^2 per GPU): encode -> decode with the encoded latents reproduces the
    HR batch (invertibility of the whole flow at full size), and the quantised-LR round trip the reference's test
    loop runs (HCFlow_Rescaling_model.py:306-324) stays finite."""
    cfg, net = _full_net("Rescaling_DF2K_4X", 1234)
    g = torch.Generator().manual_seed(79)
    hr = torch.rand(8, 3, 640, 640, generator=g).cuda()
    net.set_precision(precision)
    try:
        with torch.no_grad():
            lr_raw, z1, z2 = net.normal_flow_diracLR(hr, clamp=False)
            back = net.reverse_flow_diracLR(lr_raw, None, None, eps_std=1.0, eps=[z2, z1], clamp=False)
            assert maxdiff(back, hr) <= 5e-4
            lr_hat, _, _ = net(hr=hr, reverse=False)
            lrq = (torch.clamp(lr_hat, 0, 1) * 255.).round() / 255.
            rt = net(lr=lrq, eps_std=1.0, reverse=True)
            assert bool(torch.isfinite(rt).all()) and float(rt.min()) >= 0.0 and float(rt.max()) <= 1.0
    finally:
        net.set_precision("exact")


def test_full_size_div2k_validation_image_ragged_shape():
    """A whole DIV2K validation LR image as test_HCFlow.py feeds it (batch 1, 339 x 510 -> HR 1356 x 2040: odd sizes,
    no multiple of the 8 x 32 tile): f16x3 within 1e-4 of the exact kernels, tau = 0 reproducible, finite."""
    cfg, net = _full_net("SR_DF2K_4X", 1234)
    g = torch.Generator().manual_seed(80)
    lr = torch.rand(1, 3, 339, 510, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.9 for s in eps_shapes(cfg, 1, 339, 510)]
    with torch.no_grad():
        net.set_precision("exact")
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.9, eps=eps, clamp=False)
        assert tuple(ex.shape) == (1, 3, 1356, 2040) and bool(torch.isfinite(ex).all())
        net.set_precision("f16x3")
        try:
            fa = net.reverse_flow_diracLR(lr, None, None, eps_std=0.9, eps=eps, clamp=False)
            d0 = net(lr=lr, eps_std=0.0, reverse=True)
            d1 = net(lr=lr, eps_std=0.0, reverse=True)
        finally:
            net.set_precision("exact")
    assert maxdiff(fa, ex) <= 1e-4 * max(1.0, float(ex.abs().max()))
    assert torch.equal(d0, d1) and float(d0.min()) >= 0.0 and float(d0.max()) <= 1.0


def test_rebinding_to_new_parameter_tensors_refreshes_on_device():
    """Parameters replaced by NEW tensors of the same shapes (what nn.DataParallel's replicate does every forward, or
    ``p.data = other``): the engine re-binds and repacks on the device; results equal a freshly built module."""
    from hcflow_amd import HCFlowNet_SR
    cfg = preset("SR_4X_tiny")
    pa, pb = cached_params("SR_4X_tiny", 11), cached_params("SR_4X_tiny", 12)
    net = build_net(cfg, pa)
    g = torch.Generator().manual_seed(4)
    lr = torch.rand(2, 3, 24, 24, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.6 for s in eps_shapes(cfg, 2, 24, 24)]
    with torch.no_grad():
        net.reverse_flow_diracLR(lr, None, None, eps_std=0.6, eps=eps, clamp=False)       # host path, binds pointers
        for k, p in net.named_parameters():
            p.data = pb[k].to("cuda:0").clone()                                            # new tensors, new values
        got = net.reverse_flow_diracLR(lr, None, None, eps_std=0.6, eps=eps, clamp=False)
    assert net._engines[0]["ptrs"] is not None
    ref = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    ref.load_state_dict(pb, strict=True)
    for m in ref.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    with torch.no_grad():
        want = ref.to("cuda:0").eval().reverse_flow_diracLR(lr, None, None, eps_std=0.6, eps=eps, clamp=False)
    assert maxdiff(got, want) <= 2e-5 * max(1.0, float(want.abs().max()))
    with torch.no_grad():                                   # leave the cached module as other tests expect it
        for k, p in net.named_parameters():
            p.data = pa[k].to("cuda:0").clone()


def test_empty_batch_samples_to_an_empty_batch():
    """netG(lr=<0 images>, reverse=True): the reference's reverse path (HCFlowNet_SR_arch.py:70-75) is convs / elementwise ops /
    randn over the batch axis, all of which accept B = 0; the engine entry rejects B < 1, so the module answers itself."""
    cfg = preset("SR_4X_tiny")
    net = build_net(cfg, cached_params("SR_4X_tiny", 11))
    with torch.no_grad():
        out = net(lr=torch.zeros(0, 3, 12, 20).cuda(), eps_std=0.8, reverse=True)
    assert tuple(out.shape) == (0, 3, 48, 80) and out.dtype == torch.float32 and out.is_cuda
