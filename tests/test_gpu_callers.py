"""-m gpu: replay of the calls the REFERENCE'S OWN CALLERS made on top of the drop-in classes (recorded in the build container by
tests/golden/run_reference_callers.py, where their arithmetic was delegated to the reference module): the same keyword
arguments, the same torch random draws (regenerated from the driver's seed), through the engine -- outputs, test() return
values, training logs and post-step parameters must equal the reference's.
  test_HCFlow.py:85-90 -> HCFlowSRModel.test / HCFLowRescalingModel.test (HCFlow_SR_model.py:281-301, HCFlow_Rescaling_model.py:306-324)
  train_HCFlow.py loop   -> optimize_parameters (HCFlow_SR_model.py:184-205, HCFlow_Rescaling_model.py:204-256)"""
import numpy as np
import pytest
import torch

from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling, make_params
from tests.util import load_golden, caller_calls, caller_cfg, regen_draws, check_packed, real_inputs, maxdiff, t

pytestmark = pytest.mark.gpu


def _images(tag):
    im = load_golden("real_images")

    def cv(a):
        a = a[None] if a.ndim == 3 else a
        return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2))).float() / 255.
    if tag == "test_sr8":
        return cv(im["face_lr"][:1]), cv(im["face_hr"][:1])
    return cv(im["butterfly_lr"]), cv(im["butterfly_hr"])


@pytest.mark.parametrize("tag", ["test_sr4", "test_sr8", "test_rescale"])
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_replay_of_the_reference_test_driver(tag, precision):
    g = load_golden("callers_test")
    cfg = caller_cfg(g, tag)
    calls = caller_calls(g, tag)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, int(g[tag + "_seed"])), strict=True)     # what load_network put there
    for m in net.modules():                                                        # HCFlow_SR_model.load(): set_actnorm_init(True)
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").set_precision(precision)
    lr, hr = _images(tag)
    data = {"lr": lr.cuda(), "hr": hr.cuda()}
    draws = regen_draws(g, tag, seed=0)
    heats = [float(h) for h in g[tag + "_heats"]]
    net.eval()                                                                     # test(): self.netG.eval()
    with torch.no_grad():
        # call 0: hr -> lr (+ nll)
        kw = {k: (data[k] if isinstance(v, tuple) else v) for k, v in calls[0][4].items()}
        assert calls[0][4]["hr"][1] == tuple(hr.shape) and calls[0][4]["lr"][1] == tuple(lr.shape)
        if cfg.sr:
            kind, noise = draws.pop(0)
            assert kind == "rand"
            lq, nll = net(**kw, noise=noise.cuda())
            ret = float(nll.mean())
            assert abs(ret - float(g[tag + "_test_return"])) <= 1e-5 * abs(float(g[tag + "_test_return"]))
            assert maxdiff(lq[0], g[tag + "_lq_fromH"]) <= 1.0 / 255 + 1e-6
            lr_in = data["lr"]
        else:
            lq, z1, z2 = net(**kw)
            assert abs(float(z1.mean()) - float(g[tag + "_test_return"])) <= 1e-5
            lqq = (torch.clamp(lq, 0, 1) * 255.).round() / 255.                    # Basic.Quantization
            ref_q = t(g[tag + "_lq_fromH"]).cuda()[None]
            # levels, not bits: torch divides by a scalar as x * (1 / 255) on the GPU (1 ulp from the CPU's x / 255); a 1e-6
            # deviation before the rounding may flip a 1/255 level
            assert float(((lqq - ref_q).abs() > 0.5 / 255).float().mean()) < 0.01
            lr_in = ref_q                                                          # decode what the reference decoded
        # calls 1..: lr (+ eps) -> hr per heat / sample
        for c, heat in zip(calls[1:], [h for h in heats for _ in range(int(g[tag + "_n_sample"]))]):
            eps = []
            for _ in range(cfg.L):
                kind, e = draws.pop(0)
                assert kind == "normal"
                eps.append(e)
            kw = {k: (lr_in if isinstance(v, tuple) else v) for k, v in c[4].items()}
            out = net(**kw, eps=eps)
            check_packed(g, "%s_SR_%g_0" % (tag, heat), out[0], 1e-4)
    assert not draws
    assert net.engine().fallback_count() == 0


def _train_net(g, tag, precision):
    cfg = caller_cfg(g, tag)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    p = make_params(cfg, int(g[tag + "_seed"]))
    for k in p:
        if ".actnorm." in k:
            p[k] = torch.zeros_like(p[k])                      # a fresh training run: ActNorms still to be fitted
    net.load_state_dict(p, strict=True)
    net = net.to("cuda:0").train().set_precision(precision)
    lr_, b1, b2, wd, eps = [float(v) for v in g[tag + "_optim_hyper"]]
    params = [q for q in net.parameters() if q.requires_grad]
    opt = torch.optim.Adam(params, lr=lr_, weight_decay=wd, betas=(b1, b2), eps=eps)
    return cfg, net, opt


def _clip(net, g, tag):
    clip, norm = [float(v) for v in g[tag + "_max_grad"]]
    if clip > 0:
        torch.nn.utils.clip_grad_value_(net.parameters(), clip)
    if norm > 0:
        torch.nn.utils.clip_grad_norm_(net.parameters(), norm)


def _check_params(net, g, tag):
    want = g[tag + "_param_digest"]
    for (k, v), (n2, s1) in zip(net.state_dict().items(), want):
        assert abs(float(v.double().norm()) - n2) <= 2e-4 * max(1.0, n2), (k, float(v.double().norm()), n2)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_replay_of_optimize_parameters_sr(precision):
    """Two HCFlowSRModel.optimize_parameters steps (step < act_norm_start_step: ActNorms re-armed and fitted in step 0,
    nll.backward, gradient clip, Adam): the logged nll of both steps and every parameter afterwards equal the reference run."""
    tag = "train_sr4"
    g = load_golden("callers_train")
    cfg, net, opt = _train_net(g, tag, precision)
    hr, lr = t(g[tag + "_hr"]).cuda(), t(g[tag + "_lr"]).cuda()
    draws = regen_draws(g, tag, seed=0)
    calls = caller_calls(g, tag)
    for step, c in enumerate(calls):
        for m in net.modules():                                 # set_actnorm_init(inited=False), HCFlow_SR_model.py:186-187
            if "ActNorm" in type(m).__name__:
                m.inited = False
        kw = {k: ({"hr": hr, "lr": lr}[k] if isinstance(v, tuple) else v) for k, v in c[4].items()}
        _, nll = net(**kw, noise=draws.pop(0)[1].cuda())
        nll = float(g[tag + "_recipe"][0]) * nll.sum()
        assert abs(float(nll.detach()) - float(g[tag + "_logs"][step][0])) <= 2e-4 * abs(float(g[tag + "_logs"][step][0])), (step, float(nll.detach()))
        nll.backward()
        _clip(net, g, tag)
        opt.step()
        opt.zero_grad()
    _check_params(net, g, tag)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_replay_of_optimize_parameters_rescaling(precision):
    """Two HCFLowRescalingModel.optimize_parameters steps: forward -> LR / z losses -> Quantization -> inverse -> HR loss, one
    backward through both passes, clip, Adam; the three logged losses of both steps and the parameters equal the reference run."""
    tag = "train_rescale"
    g = load_golden("callers_train")
    cfg, net, opt = _train_net(g, tag, precision)
    hr, lr = t(g[tag + "_hr"]).cuda(), t(g[tag + "_lr"]).cuda()
    w_lr, w_z, w_hr, tau = [float(v) for v in g[tag + "_recipe"]]
    assert [str(v) for v in g[tag + "_criteria"]] == ["MSELoss", "L1Loss"]
    keys = [str(k) for k in g[tag + "_log_keys"]]
    draws = regen_draws(g, tag, seed=0)
    calls = caller_calls(g, tag)

    class Quant(torch.autograd.Function):                       # Basic.Quant (Basic.py:186-196): straight-through
        @staticmethod
        def forward(ctx, x):
            return (torch.clamp(x, 0, 1) * 255.).round() / 255.

        @staticmethod
        def backward(ctx, gout):
            return gout

    for step in range(2):
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = False
        opt.zero_grad()
        c_f, c_r = calls[2 * step], calls[2 * step + 1]
        kw = {k: ({"hr": hr, "lr": lr}[k] if isinstance(v, tuple) else v) for k, v in c_f[4].items()}
        fake_lr, z1, z2 = net(**kw)
        l_lr = w_lr * torch.nn.functional.mse_loss(fake_lr, lr)
        l_z = w_z * (torch.cat([z1.flatten(), z2.flatten()], 0) ** 2).mean()
        q = Quant.apply(fake_lr)
        eps = [draws.pop(0)[1], draws.pop(0)[1]]
        kw = {k: (q if isinstance(v, tuple) else v) for k, v in c_r[4].items()}
        fake_h = net(**kw, eps=eps)
        l_hr = w_hr * torch.nn.functional.l1_loss(fake_h, hr)
        got = {"l_g_lr": float(l_lr), "l_g_z": float(l_z), "l_g_hr": float(l_hr)}
        for i, k in enumerate(keys):
            want = float(g[tag + "_logs"][step][i])
            assert abs(got[k] - want) <= 1e-3 * abs(want) + 1e-7, (step, k, got[k], want)
        (l_lr + l_z + l_hr).backward()
        _clip(net, g, tag)
        opt.step()
    _check_params(net, g, tag)
