"""-m gpu: LU-decomposed InvertibleConv1x1 (Permutations.py:41-57 construction, :78-92 get_weight) through the drop-in
classes and the C ABI, against fixtures generated from the REFERENCE'S OWN FlowStep / InvertibleConv1x1 with
LU_decomposed=True (tests/golden/make_golden.py `lu`, class ReferenceLU): inverse, NLL forward / log-det, rescaling forward +
round trip, the gradients of l / log_s / u for the NLL step, the reverse-path step and the rescaling step, and an optimiser
loop (device-side refresh of the composed W)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hcflow_amd.config import preset, param_spec
from tests.util import load_golden, params_for, t, maxdiff, cached_params, spec_grads
from tests.test_gpu_nets import _eps

pytestmark = pytest.mark.gpu


def _net(cfg, p, train=False):
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0")
    return net.train() if train else net.eval()


def test_lu_state_dict_table_and_option():
    """Keys / order of the five LU tensors per step (parameters l, log_s, u, then the buffers p, sign_s) and the option that
    selects them (network_G.flowDownsampler.LU_decomposed)."""
    from hcflow_amd.config import NetConfig
    cfg = preset("SR_4X_tiny_LU")
    assert NetConfig.from_opt(cfg.to_opt()).lu
    net = _net(cfg, cached_params("SR_4X_tiny_LU", 91))
    keys = list(net.state_dict().keys())
    assert keys == [k for k, _, _ in param_spec(cfg)]
    i = keys.index("flow.layers.1.permute.l")
    assert keys[i:i + 5] == ["flow.layers.1.permute." + s for s in ("l", "log_s", "u", "p", "sign_s")]
    assert not any(k.endswith("permute.weight") for k in keys)
    bufs = dict(net.named_buffers())
    assert "flow.layers.1.permute.p" in bufs and "flow.layers.1.permute.sign_s" in bufs
    assert "flow.layers.1.permute.l" in dict(net.named_parameters())


@pytest.mark.parametrize("name", ["net_sr4_tiny_lu", "net_sr8_tiny_lu", "net_rescale_tiny_lu"])
def test_lu_inverse_matches_reference(name):
    g = load_golden(name)
    cfg, p = params_for(g)
    net = _net(cfg, p)
    with torch.no_grad():
        for ti in (0, 1):
            tau = float(g["inv%d_tau" % ti])
            eps = _eps(g, "inv%d" % ti)
            raw = net.reverse_flow_diracLR(t(g["lr"]).cuda(), None, None, eps_std=tau, eps=eps, clamp=False)
            scale = max(1.0, float(np.abs(g["inv%d_raw" % ti]).max()))
            assert maxdiff(raw, g["inv%d_raw" % ti]) <= 1e-4 * scale, (name, ti, maxdiff(raw, g["inv%d_raw" % ti]))
            out = net(lr=t(g["lr"]).cuda(), z=None, u=None, eps_std=tau, reverse=True, eps=eps)
            assert maxdiff(out, g["inv%d_out" % ti]) <= 1e-4


@pytest.mark.parametrize("name", ["net_sr4_tiny_lu", "net_sr8_tiny_lu"])
def test_lu_forward_nll_and_logdet_match_reference(name):
    """dlogdet = sum(log_s) * pixels (Permutations.py:84) enters the objective: latent, log-det and NLL."""
    g = load_golden(name)
    cfg, p = params_for(g)
    net = _net(cfg, p)
    with torch.no_grad():
        hr, lr, noise = t(g["hr"]).cuda(), t(g["lr"]).cuda(), t(g["fwd_noise"]).cuda()
        lr_hat, nll, logdet, z = net.normal_flow_diracLR(hr, lr, noise=noise, return_internals=True)
        assert maxdiff(z, g["fwd_z"]) <= 1e-4
        # (the per-sample output is the OBJECTIVE = flow log-det + Dirac term; the log-det itself is held through the NLLs below,
        #  whose lr := LR^ form leaves only the flow's log-det and the priors)
        _, nll_self = net(hr=hr, lr=t(g["fwd_lr"]).cuda(), reverse=False, noise=noise)
        assert abs(float(nll_self) - float(g["fwd_nll_self"])) <= 1e-4
        assert abs(float(nll) - float(g["fwd_nll"])) <= 1e-5 * abs(float(g["fwd_nll"]))


def test_lu_rescale_forward_and_roundtrip():
    g = load_golden("net_rescale_tiny_lu")
    cfg, p = params_for(g)
    net = _net(cfg, p)
    with torch.no_grad():
        lr_hat, z1, z2 = net(hr=t(g["hr"]).cuda(), reverse=False)
        assert maxdiff(lr_hat, g["fwd_lr"]) <= 1e-4
        assert maxdiff(z1, g["fwd_z1"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z1"]).max()))
        assert maxdiff(z2, g["fwd_z2"]) <= 1e-4 * max(1.0, float(np.abs(g["fwd_z2"]).max()))
        rt = net(lr=t(g["rt_lrq"]).cuda(), eps_std=1.0, reverse=True, eps=_eps(g, "rt"))
        assert maxdiff(rt, g["rt_out"]) <= 1e-4


def _lu_rows(cfg):
    return [i for i, (k, _, kind) in enumerate(param_spec(cfg)) if kind in ("lu_l", "lu_log_s", "lu_u")]


def test_lu_nll_step_gradients_match_reference():
    """d nll / d (l, log_s, u) and every other parameter of one NLL step (HCFlow_SR_model.py:195-199)."""
    from tests.test_oracle_golden import check_grads_against_fixture
    g = load_golden("grad_sr4_tiny_lu")
    cfg, p = params_for(g)
    net = _net(cfg, p, train=True)
    lr_hat, nll = net(hr=t(g["hr"]).cuda(), lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
    assert abs(float(nll.detach()) - float(g["fwd_nll"])) <= 2e-4 * max(1.0, abs(float(g["fwd_nll"])) / 100)
    nll.backward()
    grads = spec_grads(net, cfg)
    assert all(np.isfinite(x).all() for x in grads)
    rows = _lu_rows(cfg)
    assert rows and all(float(np.abs(grads[i]).max()) > 0 for i in rows)       # the factors DO receive gradients
    for k, _, kind in param_spec(cfg):
        if kind in ("lu_p", "lu_sign_s"):
            assert dict(net.named_buffers())[k].grad is None
    check_grads_against_fixture(g, grads)
    # the masks: l's gradient is strictly lower triangular, u's strictly upper (l * l_mask, u * l_mask^T)
    spec = param_spec(cfg)
    for i in rows:
        if spec[i][2] == "lu_l":
            assert float(np.abs(np.triu(grads[i])).max()) == 0.0
        if spec[i][2] == "lu_u":
            assert float(np.abs(np.tril(grads[i])).max()) == 0.0


def test_lu_reverse_path_gradients_match_reference():
    from tests.test_oracle_golden import check_grads_against_fixture, rgrad_eps
    g = load_golden("rgrad_sr4_tiny_lu")
    cfg, p = params_for(g)
    net = _net(cfg, p, train=True)
    fake = net(lr=t(g["lr"]).cuda(), z=None, u=None, eps_std=float(g["tau"]), reverse=True,
               eps=[e.cuda() for e in rgrad_eps(g)])
    assert float((fake.detach().cpu() - t(g["fake"])).abs().max()) <= 1e-4
    loss = F.l1_loss(fake, t(g["hr"]).cuda())
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5
    loss.backward()
    grads = spec_grads(net, cfg)
    assert all(np.isfinite(x).all() for x in grads)
    check_grads_against_fixture(g, grads, rtol=5e-4)


def test_lu_rescaling_step_gradients_match_reference():
    from tests.test_oracle_golden import check_grads_against_fixture, rgrad_eps, rescale_step_loss
    g = load_golden("grad_rescale_tiny_lu")
    cfg, p = params_for(g)
    net = _net(cfg, p, train=True)
    eps = [e.cuda() for e in rgrad_eps(g)]
    l_lr, l_z, l_hr, fake_lr, fake_h = rescale_step_loss(
        lambda x: net(hr=x, u=None, reverse=False),
        lambda x, e: net(lr=x, z=None, u=None, eps_std=1.0, reverse=True, eps=e),
        t(g["hr"]).cuda(), t(g["lr"]).cuda(), eps)
    assert float((fake_lr.detach().cpu() - t(g["fake_lr"])).abs().max()) <= 1e-4
    assert float((fake_h.detach().cpu() - t(g["fake_h"])).abs().max()) <= 1e-4
    (l_lr + l_z + l_hr).backward()
    grads = spec_grads(net, cfg)
    assert all(np.isfinite(x).all() for x in grads)
    check_grads_against_fixture(g, grads, rtol=5e-3, elem_rtol=3e-2)      # same allowance as the non-LU rescaling fixture


def test_lu_optimiser_steps_refresh_the_composed_weight_on_the_device():
    """Adam steps on l / log_s / u (HCFlow_SR_model.py:184-205): the engine re-composes W = P L U' from the updated device
    tensors (hcf_refresh_from_device); after the loop the module's outputs equal those of a FRESH module loaded from the
    updated state dict (host composition in hcf_finalize), and the loss fell."""
    cfg = preset("SR_4X_tiny_LU")
    net = _net(cfg, cached_params("SR_4X_tiny_LU", 91), train=True)
    g = torch.Generator().manual_seed(5)
    hr = torch.rand(2, 3, 64, 64, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(2, 3, 64, 64, generator=g).cuda()
    opt = torch.optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-3)
    l0 = net.flow.layers[1].permute.l.detach().clone()
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        opt.step()
        losses.append(float(nll.detach()))
    assert losses[-1] < losses[0]
    assert float((net.flow.layers[1].permute.l.detach() - l0).abs().max()) > 0
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    fresh = _net(cfg, sd)
    eps = [torch.randn(s, generator=g) * 0.7 for s in __import__("hcflow_amd").eps_shapes(cfg, 2, 16, 16)]
    with torch.no_grad():
        a = net(lr=lr, z=None, u=None, eps_std=0.7, reverse=True, eps=eps)
        b = fresh(lr=lr, z=None, u=None, eps_std=0.7, reverse=True, eps=eps)
        _, n1 = net(hr=hr, lr=lr, reverse=False, noise=noise)
        _, n2 = fresh(hr=hr, lr=lr, reverse=False, noise=noise)
    assert maxdiff(a, b) <= 1e-5
    assert abs(float(n1) - float(n2)) <= 1e-5 * abs(float(n2))


def test_lu_state_dict_with_other_pivoting_after_a_first_call():
    """net.load_state_dict(ckpt) AFTER the engine has run: a checkpoint's p / sign_s (the fixed buffers of the LU form,
    Permutations.py:46-55) differ from the random init's pivoting. The data pointers are unchanged and only _version moves, so
    the module must not take the device-side refresh (which re-reads parameters only) -- the outputs must equal the oracle's on
    the NEW state dict."""
    from oracle import hcflow_oracle as O
    from hcflow_amd.config import eps_shapes
    cfg = preset("SR_4X_tiny_LU")
    p_a, p_b = cached_params("SR_4X_tiny_LU", 91), cached_params("SR_4X_tiny_LU", 93)
    key = "flow.layers.1.permute.p"
    assert not torch.equal(torch.as_tensor(p_a[key]), torch.as_tensor(p_b[key])) or \
        not torch.equal(torch.as_tensor(p_a["flow.layers.1.permute.sign_s"]), torch.as_tensor(p_b["flow.layers.1.permute.sign_s"]))
    net = _net(cfg, p_a)
    g = torch.Generator().manual_seed(11)
    lr = torch.rand(2, 3, 12, 10, generator=g)
    eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 2, 12, 10)]
    with torch.no_grad():
        out_a = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.8, eps=eps, clamp=False)
        ref_a = O.sr_inverse(lr, p_a, cfg, 0.8, eps, clamp=False)
        assert maxdiff(out_a, ref_a) <= 1e-4 * max(1.0, float(ref_a.abs().max()))
        ptr = net.flow.layers[1].permute.p.data_ptr()
        net.load_state_dict(p_b, strict=True)                      # in place: same storage, new contents
        assert net.flow.layers[1].permute.p.data_ptr() == ptr
        out_b = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.8, eps=eps, clamp=False)
        ref_b = O.sr_inverse(lr, p_b, cfg, 0.8, eps, clamp=False)
        assert maxdiff(out_b, ref_b) <= 1e-4 * max(1.0, float(ref_b.abs().max())), maxdiff(out_b, ref_b)
        hr = torch.rand(2, 3, 48, 40, generator=g)
        noise = torch.rand(2, 3, 48, 40, generator=g)
        lr_ref, _ = O.sr_forward(hr, torch.zeros(2, 3, 12, 10), p_b, cfg, noise=noise)
        _, nll_ref = O.sr_forward(hr, lr_ref, p_b, cfg, noise=noise)
        _, nll = net(hr=hr.cuda(), lr=lr_ref.cuda(), reverse=False, noise=noise.cuda())
        assert abs(float(nll) - float(nll_ref)) <= 1e-4


def test_lu_oracle_parity_on_fresh_inputs_full_width():
    """HIP path vs the CPU oracle on new seeded inputs, LU in the x8 net (C = 48 steps: the widest composition)."""
    from oracle import hcflow_oracle as O
    from hcflow_amd.config import eps_shapes
    cfg = preset("SR_8X_tiny_LU")
    p = cached_params("SR_8X_tiny_LU", 92)
    net = _net(cfg, p)
    g = torch.Generator().manual_seed(77)
    lr = torch.rand(3, 3, 7, 9, generator=g)
    eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 3, 7, 9)]
    with torch.no_grad():
        ref = O.sr_inverse(lr, p, cfg, 0.8, eps, clamp=False)
        out = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=0.8, eps=eps, clamp=False)
    assert maxdiff(out, ref) <= 1e-4 * max(1.0, float(ref.abs().max()))
