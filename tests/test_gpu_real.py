"""-m gpu: the engine against reference-generated fixtures at FULL DEPTH and multi-tile sizes -- the reference's bundled example
images (datasets/example_general_4X, example_face_8X) with ActNorms fitted by the reference's own data-dependent init, and a
ragged 24 x 72 LR -- plus reference-written checkpoints (nn.DataParallel state dict, 'module.' keys; base_model.py:79-120).
Both conv precisions; the f16x3 range guard must stay silent on these realistic (unit-variance) activations."""
import collections

import numpy as np
import pytest
import torch

from tests.util import load_golden, real_inputs, real_params, params_for, seeded_eps, check_packed, maxdiff, t

pytestmark = pytest.mark.gpu

REAL = ["net_sr4_full_ragged", "net_rescale_full_ragged", "net_sr4_real", "net_sr8_real", "net_rescale_real"]


def _module(cfg, p, train=False):
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    if not train:
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
    net = net.to("cuda:0")
    return net.train() if train else net.eval()


@pytest.mark.parametrize("name", REAL)
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_full_depth_real_images_and_ragged_sizes_match_reference(name, precision):
    g = load_golden(name)
    cfg, p = real_params(g)
    lr, hr = real_inputs(g)
    B, _, h, w = lr.shape
    net = _module(cfg, p).set_precision(precision)
    with torch.no_grad():
        for ti in (0, 1):
            tau = float(g["inv%d_tau" % ti])
            eps = seeded_eps(cfg, B, h, w, tau, int(g["inv%d_eps_seed" % ti]))
            raw = net.reverse_flow_diracLR(lr.cuda(), None, None, eps_std=tau, eps=eps, clamp=False)
            scale = max(1.0, float(np.abs(g["inv%d_raw_sub" % ti]).max()))
            check_packed(g, "inv%d_raw" % ti, raw, 1e-4 * scale)
        if cfg.sr:
            noise = torch.rand(hr.shape, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
            lr_hat, nll = net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())
            assert abs(float(nll) - float(g["fwd_nll"])) <= 1e-5 * abs(float(g["fwd_nll"]))
            _, nll_self = net(hr=hr.cuda(), lr=t(g["fwd_lr"]).cuda(), reverse=False, noise=noise.cuda())
            assert abs(float(nll_self) - float(g["fwd_nll_self"])) <= 1e-4          # bits/dim (BASELINE.json)
            d = (lr_hat.cpu() - t(g["fwd_lr"])).abs()
            assert float(d.max()) <= 1.0 / 255 + 1e-6 and float((d > 1e-6).float().mean()) < 0.01
        else:
            lr_hat, z1, z2 = net(hr=hr.cuda(), reverse=False)
            assert maxdiff(lr_hat, g["fwd_lr"]) <= 1e-4
            check_packed(g, "fwd_z1", z1, 1e-4 * max(1.0, float(np.abs(g["fwd_z1_sub"]).max())))
            check_packed(g, "fwd_z2", z2, 1e-4 * max(1.0, float(np.abs(g["fwd_z2_sub"]).max())))
    assert net.engine().fallback_count() == 0, "f16x3 range fallback on reference-fitted weights / real images"


@pytest.mark.parametrize("name", ["net_sr4_real", "net_sr8_real", "net_rescale_real"])
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_actnorm_data_init_on_real_images_matches_reference(name, precision):
    """train() mode, ActNorms zeroed: ONE forward pass on the reference's example images fits what the reference fitted
    (full depth: 156 / 234 / 52 ActNorms, ActNorms.py:29-43)."""
    g = load_golden(name)
    cfg, p = params_for(g)
    lr, hr = real_inputs(g)
    keys = [str(k) for k in g["an_keys"]]
    p = dict(p)
    for k in keys:
        p[k + ".bias"] = torch.zeros_like(p[k + ".bias"])
        p[k + ".logs"] = torch.zeros_like(p[k + ".logs"])
    net = _module(cfg, p, train=True).set_precision(precision)
    mods = dict(net.named_modules())
    with torch.no_grad():
        noise = torch.rand(hr.shape, generator=torch.Generator().manual_seed(int(g["noise_seed"])))
        if cfg.sr:
            net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())
        else:
            net(hr=hr.cuda(), reverse=False)
    for i, k in enumerate(keys):
        assert mods[k].inited is True
        assert maxdiff(mods[k].bias.reshape(-1), g["an_bias_%d" % i]) <= 2e-4, (k, "bias")
        assert maxdiff(mods[k].logs.reshape(-1), g["an_logs_%d" % i]) <= 2e-4, (k, "logs")


def _load_network(load_net, network, strict=True):        # base_model.py:97-120 on an in-memory state dict
    if isinstance(network, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        network = network.module
    clean = collections.OrderedDict()
    for k, v in load_net.items():
        clean[k[7:] if k.startswith("module.") else k] = v
    network.load_state_dict(clean, strict=strict)


@pytest.mark.parametrize("name", ["ckpt_sr4_micro", "ckpt_rescale_micro"])
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_reference_written_checkpoint_loads_strictly_and_reproduces_reference_output(name, precision):
    """A state dict written by the REFERENCE module under nn.DataParallel ('module.' keys, reference-initialised weights)
    goes through load_network's prefix stripping into our class with strict=True and reproduces the reference's outputs."""
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    from hcflow_amd.config import preset
    g = load_golden(name)
    cfg = preset(str(g["preset"]))
    sd = collections.OrderedDict((str(k), t(g["t_%d" % i])) for i, k in enumerate(g["keys"]))
    assert all(k.startswith("module.") for k in sd)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    _load_network(sd, torch.nn.DataParallel(net), strict=True)
    for m in net.modules():                               # HCFlow_SR_model.load(): set_actnorm_init(inited=True)
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval().set_precision(precision)
    lr, hr = t(g["lr"]), t(g["hr"])
    B, _, h, w = lr.shape
    with torch.no_grad():
        eps = seeded_eps(cfg, B, h, w, 0.8, int(g["eps_seed"]))
        out = net(lr=lr.cuda(), eps_std=0.8, reverse=True, eps=eps)
        assert maxdiff(out, g["inv_out"]) <= 1e-4
        if cfg.sr:
            _, nll = net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
            assert abs(float(nll) - float(g["fwd_nll"])) <= 1e-5 * abs(float(g["fwd_nll"]))
        else:
            lr_hat, _, _ = net(hr=hr.cuda(), reverse=False)
            assert maxdiff(lr_hat, g["fwd_lr"]) <= 1e-4
    # and back: what save_network would write from our module equals what the reference wrote (keys, order, values)
    back = net.state_dict()
    assert list(back.keys()) == [k[7:] for k in sd.keys()]
    assert all(torch.equal(back[k[7:]].cpu(), v) for k, v in sd.items())
