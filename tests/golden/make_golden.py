#!/usr/bin/env python
"""Generate golden fixtures from the REFERENCE ITSELF (runs only in the build container).

    python tests/golden/make_golden.py          # writes tests/golden/*.npz

Imports the reference's arch modules from /root/reference/codes (read-only, never copied) with a
tiny stub for ``utils.util`` (the real one needs cv2/natsort/torchvision), loads
the seeded parameter recipe (hcflow_amd.params.make_params) into the reference modules with
``load_state_dict(strict=True)`` -- which also pins our state-dict key/shape table -- and records
inputs, captured random draws and outputs. Fixtures are DATA only (inputs / expected outputs);
the GPU box regenerates the weights from the same recipe and checks ``digest``.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/codes"

from hcflow_amd.config import NetConfig, preset, param_spec, eps_shapes  # noqa: E402
from hcflow_amd.params import make_params, param_digest  # noqa: E402


def import_reference():
    from hcflow_amd.config import opt_get
    utils = types.ModuleType("utils")
    util = types.ModuleType("utils.util")
    util.opt_get = opt_get
    util.register_hook = lambda *a, **k: None
    util.trunc_normal_ = lambda *a, **k: None
    utils.util = util
    sys.modules["utils"] = utils
    sys.modules["utils.util"] = util
    sys.path.insert(0, REF)
    from models.modules.HCFlowNet_SR_arch import HCFlowNet_SR
    from models.modules.HCFlowNet_Rescaling_arch import HCFlowNet_Rescaling
    return HCFlowNet_SR, HCFlowNet_Rescaling


class Capture:
    """Record (or replay) torch.normal / torch.rand draws made inside the reference."""

    def __init__(self, replay_normal=None, replay_rand=None):
        self.normal, self.rand = [], []
        self.replay_normal = list(replay_normal) if replay_normal is not None else None
        self.replay_rand = list(replay_rand) if replay_rand is not None else None

    def __enter__(self):
        self._n, self._r = torch.normal, torch.rand

        def normal(*a, **k):
            out = self.replay_normal.pop(0) if self.replay_normal is not None else self._n(*a, **k)
            self.normal.append(out.clone())
            return out

        def rand(*a, **k):
            out = self.replay_rand.pop(0) if self.replay_rand is not None else self._r(*a, **k)
            self.rand.append(out.clone())
            return out

        torch.normal, torch.rand = normal, rand
        return self

    def __exit__(self, *exc):
        torch.normal, torch.rand = self._n, self._r


class ReferenceLU:
    """The reference's FlowNets never pass ``LU_decomposed`` to FlowStep (FlowStep.py:9-10,20), so no yml can switch the
    LU-decomposed InvertibleConv1x1 (Permutations.py:41-57) on. For the LU fixtures the DEFAULT of that constructor argument
    is flipped while the reference net is built -- the reference's own FlowStep / InvertibleConv1x1 code then runs unchanged."""

    def __enter__(self):
        from models.modules import FlowStep as FS
        self.fn = FS.FlowStep.__init__
        self.old = self.fn.__defaults__
        names = self.fn.__code__.co_varnames[:self.fn.__code__.co_argcount]
        i = names.index("LU_decomposed") - (len(names) - len(self.old))
        assert self.old[i] is False
        self.fn.__defaults__ = self.old[:i] + (True,) + self.old[i + 1:]
        return self

    def __exit__(self, *exc):
        self.fn.__defaults__ = self.old


def build(ref_cls, cfg: NetConfig, seed: int):
    opt = cfg.to_opt()
    torch.manual_seed(0)
    np.random.seed(0)
    if cfg.lu:
        with ReferenceLU():
            net = ref_cls(opt=opt, step=0)
        assert any(k.endswith(".permute.log_s") for k in net.state_dict())
    else:
        net = ref_cls(opt=opt, step=0)
    params = make_params(cfg, seed)
    sd = net.state_dict()
    assert list(sd.keys()) == [k for k, _, _ in param_spec(cfg)], "state_dict key ORDER mismatch"
    for k, shape, _ in param_spec(cfg):
        assert tuple(sd[k].shape) == tuple(shape), (k, sd[k].shape, shape)
    net.load_state_dict(params, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net.eval()
    return net, params


def np_(t):
    return t.detach().cpu().numpy()


def gen_net_fixture(name, preset_name, ref_sr, ref_rs, B, h, w, seed, taus=(0.0, 0.8)):
    cfg = preset(preset_name)
    net, params = build(ref_sr if cfg.sr else ref_rs, cfg, seed)
    g = torch.Generator().manual_seed(seed + 17)
    lr = torch.rand(B, 3, h, w, generator=g)
    hr = torch.rand(B, 3, h * cfg.scale, w * cfg.scale, generator=g)
    out = {"preset": preset_name, "seed": seed, "lr": np_(lr), "hr": np_(hr)}
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    with torch.no_grad():
        for ti, tau in enumerate(taus):
            with Capture() as cap:
                y = net(lr=lr, eps_std=tau, reverse=True)
            eps = cap.normal
            assert [tuple(e.shape) for e in eps] == eps_shapes(cfg, B, h, w), [e.shape for e in eps]
            with Capture(replay_normal=eps):
                y_raw = net.flow(z=lr, eps_std=tau, reverse=True)
            assert torch.equal(torch.clamp(y_raw, 0, 1), y)
            out["inv%d_tau" % ti] = np.float64(tau)
            for i, e in enumerate(eps):
                out["inv%d_eps%d" % (ti, i)] = np_(e)
            out["inv%d_out" % ti] = np_(y)
            out["inv%d_raw" % ti] = np_(y_raw)
            frac = float(((y_raw < 0) | (y_raw > 1)).float().mean())
            print("  %s tau=%.1f raw range [%.3f, %.3f] clamped frac %.3f finite %s" % (
                name, tau, float(y_raw.min()), float(y_raw.max()), frac, bool(torch.isfinite(y_raw).all())))
            assert torch.isfinite(y_raw).all()
        if cfg.sr:
            with Capture() as cap:
                lr_hat, nll = net(hr=hr, lr=lr, reverse=False)
            noise = cap.rand[0]
            # pre-quantisation latent + logdet through the flow (HCFlowNet_SR_arch.py:52-56)
            pixels = hr.shape[2] * hr.shape[3]
            x = hr + noise / net.quant
            logdet0 = torch.zeros_like(hr[:, 0, 0, 0]) + float(-np.log(net.quant) * pixels)
            z, logdet = net.flow(hr=x, u=None, logdet=logdet0, reverse=False, training=True)
            out.update(fwd_noise=np_(noise), fwd_lr=np_(lr_hat), fwd_nll=np.float64(float(nll)),
                       fwd_z=np_(z), fwd_logdet=np_(logdet))
            # self-consistent case: lr := LR^ (what a trained net sees), same noise replayed
            with Capture(replay_rand=[noise]):
                lr_hat2, nll_self = net(hr=hr, lr=lr_hat, reverse=False)
            assert torch.equal(lr_hat2, lr_hat)
            out.update(fwd_nll_self=np.float64(float(nll_self)))
            print("  %s nll %.6f nll_self %.6f logdet %s" % (name, float(nll), float(nll_self), np_(logdet)))
            assert np.isfinite(float(nll)) and np.isfinite(float(nll_self))
        else:
            lr_hat, z1, z2 = net(hr=hr, reverse=False)
            zraw, _, _ = net.flow(hr=hr, u=None, logdet=None, reverse=False)
            out.update(fwd_lr=np_(lr_hat), fwd_z1=np_(z1), fwd_z2=np_(z2), fwd_raw=np_(zraw))
            # round trip as HCFLowRescalingModel.test does (HCFlow_Rescaling_model.py:306-324)
            from models.modules import Basic
            lrq = Basic.Quantization()(lr_hat)
            with Capture() as cap:
                rt = net(lr=lrq, eps_std=1.0, reverse=True)
            for i, e in enumerate(cap.normal):
                out["rt_eps%d" % i] = np_(e)
            out.update(rt_lrq=np_(lrq), rt_out=np_(rt))
            print("  %s round-trip PSNR-ish mse %.3e" % (name, float(((rt - hr) ** 2).mean())))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_aninit_fixture(name, preset_name, ref_sr, ref_rs, B, h, w, seed):
    """ActNorm data-dependent initialisation (ActNorms.py:29-43): our seeded weights with every ActNorm's bias / logs
    zeroed and ``inited = False``, ONE forward pass in train() mode; records the parameters the reference fits and
    the outputs of that same pass."""
    cfg = preset(preset_name)
    net, params = build(ref_sr if cfg.sr else ref_rs, cfg, seed)
    an = [(k, m) for k, m in net.named_modules() if "ActNorm" in type(m).__name__]
    for _, m in an:
        m.bias.data.zero_()
        m.logs.data.zero_()
        m.inited = False
    net.train()
    g = torch.Generator().manual_seed(seed + 29)
    hr = torch.rand(B, 3, h * cfg.scale, w * cfg.scale, generator=g)
    lr = torch.rand(B, 3, h, w, generator=g)
    out = {"preset": preset_name, "seed": seed, "hr": np_(hr), "lr": np_(lr)}
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    with torch.no_grad():
        if cfg.sr:
            with Capture() as cap:
                lr_hat, nll = net(hr=hr, lr=lr, reverse=False)
            out.update(fwd_noise=np_(cap.rand[0]), fwd_lr=np_(lr_hat), fwd_nll=np.float64(float(nll)))
            print("  %s nll after init %.6f" % (name, float(nll)))
        else:
            lr_hat, z1, z2 = net(hr=hr, reverse=False)
            out.update(fwd_lr=np_(lr_hat), fwd_z1=np_(z1), fwd_z2=np_(z2))
    assert all(m.inited for _, m in an)
    out["an_keys"] = np.array([k for k, _ in an])
    for i, (k, m) in enumerate(an):
        out["an_bias_%d" % i] = np_(m.bias).reshape(-1)
        out["an_logs_%d" % i] = np_(m.logs).reshape(-1)
    print("  %s: %d ActNorms fitted, logs range [%.3f, %.3f]" % (
        name, len(an), min(float(m.logs.min()) for _, m in an), max(float(m.logs.max()) for _, m in an)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def spec_tensors(net, cfg):
    """(key, tensor) of the reference net in param_spec (= state_dict) order: parameters with their .grad, buffers as they are."""
    prm = dict(net.named_parameters())
    buf = dict(net.named_buffers())
    return [(k, prm[k] if k in prm else buf[k]) for k, _, _ in param_spec(cfg)]


def grad_digest(g, i):
    """Three numbers per gradient tensor: l2 norm, sum, and a seeded random projection."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    r = np.random.RandomState(1000 + i).standard_normal(g.size)
    return float(np.sqrt((g * g).sum())), float(g.sum()), float((g * r).sum())


def gen_grad_fixture(name, preset_name, ref_sr, B, h, w, seed):
    """d nll / d parameters of ONE reference NLL step (HCFlow_SR_model.py:195-199: nll of netG(hr, lr), backward)
    on our seeded weights; digests for every tensor, full gradients for the small ones."""
    cfg = preset(preset_name)
    net, params = build(ref_sr, cfg, seed)
    net.train()                                   # ActNorms are marked inited: train() == eval() arithmetic
    g = torch.Generator().manual_seed(seed + 41)
    hr = torch.rand(B, 3, h * cfg.scale, w * cfg.scale, generator=g)
    lr = torch.rand(B, 3, h, w, generator=g)
    with Capture() as cap:
        lr_hat, nll = net(hr=hr, lr=lr, reverse=False)
    nll.backward()
    out = {"preset": preset_name, "seed": seed, "hr": np_(hr), "lr": np_(lr), "fwd_noise": np_(cap.rand[0]),
           "fwd_nll": np.float64(float(nll)), "fwd_lr": np_(lr_hat)}
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    dig, nfull = [], 0
    for i, (k, prm) in enumerate(spec_tensors(net, cfg)):     # state-dict order; buffers (LU p / sign_s) carry zero gradients
        gr = np.zeros(tuple(prm.shape), np.float32) if getattr(prm, "grad", None) is None else np_(prm.grad)
        dig.append(grad_digest(gr, i))
        if gr.size <= 2304:
            out["g_%d" % i] = gr
            nfull += 1
    keys = [k for k, _ in net.named_parameters()]
    assert keys == [k for k, _, kind in param_spec(cfg) if kind not in ("lu_p", "lu_sign_s")]
    out["gdigest"] = np.array(dig, dtype=np.float64)
    print("  %s nll %.6f: %d tensors, %d stored in full, |g| range [%.3e, %.3e]" % (
        name, float(nll), len(dig), nfull, min(d[0] for d in dig), max(d[0] for d in dig)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_rgrad_fixture(name, preset_name, ref_sr, B, h, w, seed, tau):
    """Gradients through the REVERSE path (HCFlow_SR_model.optimize_parameters :207-216, the HR pixel loss of the
    HCFlow+ / ++ recipes): fake_H = netG(lr, eps_std, reverse=True); L1(fake_H, real_H).backward(). The eps draws are
    captured so that the same sample can be replayed."""
    cfg = preset(preset_name)
    net, params = build(ref_sr, cfg, seed)
    net.train()
    g = torch.Generator().manual_seed(seed + 43)
    lr = torch.rand(B, 3, h, w, generator=g)
    hr = torch.rand(B, 3, h * cfg.scale, w * cfg.scale, generator=g) * 0.6 + 0.2
    with Capture() as cap:
        fake = net(lr=lr, z=None, u=None, eps_std=tau, reverse=True)
    loss = torch.nn.functional.l1_loss(fake, hr)
    loss.backward()
    out = {"preset": preset_name, "seed": seed, "lr": np_(lr), "hr": np_(hr), "tau": np.float64(tau),
           "fake": np_(fake), "loss": np.float64(float(loss.detach()))}
    for i, e in enumerate(cap.normal):
        out["eps%d" % i] = np_(e)
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    dig, nfull = [], 0
    for i, (k, prm) in enumerate(spec_tensors(net, cfg)):     # state-dict order; buffers (LU p / sign_s) carry zero gradients
        gr = np.zeros(tuple(prm.shape), np.float32) if getattr(prm, "grad", None) is None else np_(prm.grad)
        dig.append(grad_digest(gr, i))
        if gr.size <= 2304:
            out["g_%d" % i] = gr
            nfull += 1
    out["gdigest"] = np.array(dig, dtype=np.float64)
    frac = float(((fake <= 0) | (fake >= 1)).float().mean())
    print("  %s loss %.6f clamped %.3f: %d tensors, %d in full, |g| range [%.3e, %.3e]" % (
        name, float(loss.detach()), frac, len(dig), nfull, min(d[0] for d in dig), max(d[0] for d in dig)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_metrics_fixture():
    """Validation metrics: the reference's imresize (pure numpy), calculate_psnr and bgr2ycbcr run here as they are. Its SSIM
    (utils/util.py:914-955) calls two OpenCV primitives and this image has no OpenCV: the reference's OWN ssim / calculate_ssim /
    calculate_psnr_ssim are run against a stand-in `cv2` module that implements exactly those two from OpenCV's documentation --
    getGaussianKernel(ksize, sigma > 0) = exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)) normalised to sum 1, as a ksize x 1 float64
    column; filter2D(src, -1, kernel) = correlation, anchor at the kernel centre, BORDER_REFLECT_101 (scipy.ndimage.correlate,
    mode="mirror") -- so everything else of the function (constants, crop, maps, means, channel and Y handling, crop_border) is
    the reference's code, and the fixture also records that the border rule cannot matter (`ssim_border_dependence_*`: the
    values with a zero border instead)."""
    import types
    import importlib.util as ilu
    spec0 = ilu.spec_from_file_location("ref_utils_imresize", os.path.join(REF, "utils", "imresize.py"))
    im = ilu.module_from_spec(spec0)
    spec0.loader.exec_module(im)
    imresize = im.imresize
    for name in ("cv2", "natsort", "torchvision", "torchvision.utils"):       # only imported at module level
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision.utils"].make_grid = None
    real_util = sys.modules.pop("utils.util", None)                            # drop the opt_get stub, load the real file
    spec = ilu.spec_from_file_location("ref_utils_util", os.path.join(REF, "utils", "util.py"))
    ru = ilu.module_from_spec(spec)
    try:
        spec.loader.exec_module(ru)
    finally:
        if real_util is not None:
            sys.modules["utils.util"] = real_util
    spec2 = ilu.spec_from_file_location("ref_data_util", os.path.join(REF, "data", "util.py"))
    du = ilu.module_from_spec(spec2)
    spec2.loader.exec_module(du)
    from scipy import ndimage
    cv2s = sys.modules["cv2"]

    def _gk(ksize, sigma):
        i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(i * i) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)
    border = {"mode": "mirror"}
    cv2s.getGaussianKernel = _gk
    cv2s.filter2D = lambda src, ddepth, kernel: ndimage.correlate(np.asarray(src, dtype=np.float64), kernel, **border)
    ru.cv2 = cv2s
    rs = np.random.RandomState(7)
    out = {}
    for tag, (h, w) in {"a": (48, 64), "b": (37, 53)}.items():
        gt = rs.rand(3, h, w).astype(np.float32)
        sr = np.clip(gt + rs.randn(3, h, w).astype(np.float32) * 0.05, -0.1, 1.1)
        out["gt_" + tag], out["sr_" + tag] = gt, sr
        g8 = np.transpose(np.clip(gt, 0, 1)[[2, 1, 0]], (1, 2, 0))
        s8 = np.transpose(np.clip(sr, 0, 1)[[2, 1, 0]], (1, 2, 0))
        g8 = (g8 * 255.0).round().astype(np.uint8) / 255.        # tensor2img(...) / 255.  (uint8 -> float64, test_HCFlow.py:139-142)
        s8 = (s8 * 255.0).round().astype(np.uint8) / 255.
        out["psnr_" + tag] = np.float64(ru.calculate_psnr(g8 * 255, s8 * 255))
        out["y_" + tag] = du.bgr2ycbcr(g8.copy(), only_y=True)
        out["psnr_y_" + tag] = np.float64(ru.calculate_psnr(du.bgr2ycbcr(g8.copy(), only_y=True) * 255,
                                                            du.bgr2ycbcr(s8.copy(), only_y=True) * 255))
        out["down4_" + tag] = imresize(g8.astype(np.float64), 0.25)
        out["down2_" + tag] = imresize(s8.astype(np.float64), 0.5)
        # SSIM by the reference's functions over the documented cv2 stand-in (see the docstring)
        border.update(mode="mirror")
        out["ssim_" + tag] = np.float64(ru.calculate_ssim(g8 * 255, s8 * 255))
        for cb in (0, 4):     # (copies: the reference's bgr2ycbcr scales a float input IN PLACE, data/util.py:209-230)
            out["psnr_ssim_cb%d_%s" % (cb, tag)] = np.asarray(ru.calculate_psnr_ssim(g8.copy(), s8.copy(), crop_border=cb), dtype=np.float64)
        border.update(mode="constant", cval=0.0)
        out["ssim_border_dependence_" + tag] = np.float64(abs(ru.calculate_ssim(g8 * 255, s8 * 255) - float(out["ssim_" + tag])))
        border.pop("cval")
        border.update(mode="mirror")
    path = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_rescale_grad_fixture(name, preset_name, ref_rs, B, h, w, seed):
    """One generator step of HCFlow_Rescaling_model.optimize_parameters (:212-238) with the shipped weights of
    train_Rescaling_DF2K_4X_HCFlow.yml (:91-98): l = 5e-2 MSE(fake_LR, LR) + 1e-5 mean(z^2) + L1(fake_H, HR), where
    fake_H = netG(lr=Quant(fake_LR), eps_std=1.0, reverse=True); the eps draws are captured."""
    from models.modules.Basic import Quantization
    cfg = preset(preset_name)
    net, params = build(ref_rs, cfg, seed)
    net.train()
    g = torch.Generator().manual_seed(seed + 47)
    hr = torch.rand(B, 3, h * 4, w * 4, generator=g) * 0.8 + 0.1
    lr = torch.nn.functional.avg_pool2d(hr, 4)
    fake_lr, z1, z2 = net(hr=hr, lr=lr, u=None, reverse=False)
    l_lr = 5e-2 * torch.nn.functional.mse_loss(fake_lr, lr)
    l_z = 1e-5 * (torch.cat([z1.flatten(), z2.flatten()], 0) ** 2).mean()
    q = Quantization()(fake_lr)
    with Capture() as cap:
        fake_h = net(lr=q, z=None, u=None, eps_std=1.0, reverse=True)
    l_hr = 1.0 * torch.nn.functional.l1_loss(fake_h, hr)
    total = l_lr + l_z + l_hr
    total.backward()
    out = {"preset": preset_name, "seed": seed, "hr": np_(hr), "lr": np_(lr), "fake_lr": np_(fake_lr), "z1": np_(z1),
           "z2": np_(z2), "fake_h": np_(fake_h), "l_lr": np.float64(float(l_lr.detach())),
           "l_z": np.float64(float(l_z.detach())), "l_hr": np.float64(float(l_hr.detach()))}
    for i, e in enumerate(cap.normal):
        out["eps%d" % i] = np_(e)
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    dig, nfull = [], 0
    for i, (k, prm) in enumerate(spec_tensors(net, cfg)):     # state-dict order; buffers (LU p / sign_s) carry zero gradients
        gr = np.zeros(tuple(prm.shape), np.float32) if getattr(prm, "grad", None) is None else np_(prm.grad)
        dig.append(grad_digest(gr, i))
        if gr.size <= 2304:
            out["g_%d" % i] = gr
            nfull += 1
    out["gdigest"] = np.array(dig, dtype=np.float64)
    print("  %s l_lr %.3e l_z %.3e l_hr %.4f: %d tensors, %d in full, |g| range [%.3e, %.3e]" % (
        name, float(l_lr.detach()), float(l_z.detach()), float(l_hr.detach()), len(dig), nfull,
        min(d[0] for d in dig), max(d[0] for d in dig)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_op_fixture(ref_sr, ref_rs):
    """Per-op pins straight from the reference's modules (SURVEY.md 8c 'Per-op pins')."""
    from models.modules import Basic, thops
    from models.modules.FlowStep import FlowStep
    from models.modules.ConditionalFlow import ConditionalFlow
    out = {}
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 8, 12, generator=g)
    out["sq_in"] = np_(x)
    out["sq_out"] = np_(Basic.squeeze2d(x, 2))
    y = torch.randn(2, 12, 4, 6, generator=g)
    out["usq_in"] = np_(y)
    out["usq_out"] = np_(Basic.unsqueeze2d(y, 2))
    haar = Basic.HaarDownsampling(3)
    with torch.no_grad():
        hf, _ = haar(x, reverse=False)
        hi, _ = haar(hf, reverse=True)
        y12 = torch.randn(2, 12, 4, 6, generator=g)
        hi2, _ = haar(y12, reverse=True)
    out.update(haar_in=np_(x), haar_fwd=np_(hf), haar_inv_of_fwd=np_(hi), haar_inv_in=np_(y12),
               haar_inv_out=np_(hi2))
    # split_feature on odd C
    z21 = torch.randn(2, 21, 4, 4, generator=g)
    a, b = thops.split_feature(z21, "split")
    c, d = thops.split_feature(z21, "cross")
    out.update(split_in=np_(z21), split_a=np_(a), split_b=np_(b), cross_a=np_(c), cross_b=np_(d))
    # GaussianDiag
    mean = torch.randn(2, 6, 4, 4, generator=g)
    logs = torch.randn(2, 6, 4, 4, generator=g) * 0.3
    xx = torch.randn(2, 6, 4, 4, generator=g)
    out.update(g_mean=np_(mean), g_logs=np_(logs), g_x=np_(xx),
               g_logp=np_(Basic.GaussianDiag.logp(mean, logs, xx)))
    # Quant
    q = torch.randn(2, 3, 5, 5, generator=g) * 0.7 + 0.5
    out.update(q_in=np_(q), q_out=np_(Basic.Quantization()(q)))
    # ActNorm data init (training mode, fresh module)
    from models.modules.ActNorms import ActNorm2d
    an = ActNorm2d(5)
    an.train()
    xi = torch.randn(4, 5, 6, 6, generator=g) * 2 + 1
    with torch.no_grad():
        yo, _ = an(xi)
    out.update(ani_in=np_(xi), ani_bias=np_(an.bias), ani_logs=np_(an.logs), ani_out=np_(yo))
    np.savez_compressed(os.path.join(HERE, "ops_index.npz"), **out)

    # FlowStep / ConditionalFlow pins use the seeded recipe on a tiny SR net so that the
    # parameters come from make_params (regenerable) -- take sub-modules of the loaded net.
    cfg = preset("SR_4X_tiny")
    net, params = build(ref_sr, cfg, 4321)
    o2 = {"preset": "SR_4X_tiny", "seed": 4321}
    with torch.no_grad():
        # unconditional FlowStep at level 0: flow.layers.1 (C=12)
        fs = net.flow.layers[1]
        z = torch.randn(2, 12, 6, 10, generator=g)
        ld0 = torch.zeros(2)
        zf, ld = fs(z, None, logdet=ld0.clone(), reverse=False)
        zi, _ = fs(zf, None, reverse=True)
        o2.update(fs_in=np_(z), fs_fwd=np_(zf), fs_logdet=np_(ld), fs_inv_of_fwd=np_(zi))
        # FCN inside it
        f = fs.affine.f
        z1 = torch.randn(2, 6, 6, 10, generator=g)
        o2.update(fcn_in=np_(z1), fcn_out=np_(f(z1)))
        # conditional FlowStep of level1_condFlow (C=21, cond 128)
        cfl = net.flow.level1_condFlow
        cs = cfl.additional_flow_steps[0]
        zc = torch.randn(2, 21, 5, 7, generator=g)
        u = torch.randn(2, 128, 5, 7, generator=g) * 0.5
        zcf, ldc = cs(zc, u, logdet=torch.zeros(2), reverse=False)
        zci, _ = cs(zcf, u, reverse=True)
        o2.update(cs_in=np_(zc), cs_u=np_(u), cs_fwd=np_(zcf), cs_logdet=np_(ldc), cs_inv_of_fwd=np_(zci))
        # RDB / RRDB / cond features
        r = cfl.RRDB_trunk0[0]
        xr = torch.randn(2, 64, 5, 7, generator=g) * 0.5
        o2.update(rrdb_in=np_(xr), rdb_out=np_(r.RDB1(xr)), rrdb_out=np_(r(xr)))
        lr = torch.rand(2, 3, 5, 7, generator=g)
        o2.update(cf_in=np_(lr), cf_out=np_(cfl.get_conditional_feature_SR(lr)))
        # prior head
        h = cfl.f(u)
        o2.update(head_out=np_(h))
    np.savez_compressed(os.path.join(HERE, "ops_sr_tiny.npz"), **o2)

    cfg = preset("Rescaling_4X_tiny")
    net, params = build(ref_rs, cfg, 4322)
    o3 = {"preset": "Rescaling_4X_tiny", "seed": 4322}
    with torch.no_grad():
        for name, idx in (("even", 1), ("odd", 2)):       # LRvsothers True / False
            fs = net.flow.layers[idx]
            z = torch.randn(2, 12, 6, 10, generator=g)
            zf, _ = fs(z, None, reverse=False)
            zi, _ = fs(zf, None, reverse=True)
            o3.update({"fs_%s_in" % name: np_(z), "fs_%s_fwd" % name: np_(zf),
                       "fs_%s_inv_of_fwd" % name: np_(zi)})
            db = fs.affine.f
            cin = db.conv1.weight.shape[1]
            xd = torch.randn(2, cin, 6, 10, generator=g)
            o3.update({"db_%s_in" % name: np_(xd), "db_%s_out" % name: np_(db(xd))})
        cfl = net.flow.level1_condFlow
        lr = torch.rand(2, 3, 5, 7, generator=g)
        o3.update(cf_in=np_(lr), cf_out=np_(cfl.get_conditional_feature_Rescaling(lr)))
    np.savez_compressed(os.path.join(HERE, "ops_rescaling_tiny.npz"), **o3)
    print("wrote op fixtures")



# ------------------------------------------------------------------ fixtures on the reference's bundled example images
IMG_ROOT = "/root/reference/datasets"


def gen_real_images_fixture():
    """The example images the reference ships for its own test configs (datasets/example_general_4X, example_face_8X;
    test_SR_DF2K_4X_HCFlow.yml:13-22, test_SR_CelebA_8X_HCFlow.yml:13-22) as uint8 RGB arrays: DATA the GPU box needs as
    inputs (the reference tree does not travel)."""
    from PIL import Image

    def rd(path):
        return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    out = {"butterfly_lr": rd(IMG_ROOT + "/example_general_4X/LR/butterfly.png"),
           "butterfly_hr": rd(IMG_ROOT + "/example_general_4X/HR/butterfly.png")}
    names = sorted(os.listdir(IMG_ROOT + "/example_face_8X/LR"), key=lambda n: int(os.path.splitext(n)[0]))
    out["face_names"] = np.array(names)
    out["face_lr"] = np.stack([rd(IMG_ROOT + "/example_face_8X/LR/" + n) for n in names])
    out["face_hr"] = np.stack([rd(IMG_ROOT + "/example_face_8X/HR/" + n) for n in names])
    path = os.path.join(HERE, "real_images.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), {k: v.shape for k, v in out.items()})


def img_tensor(a):
    """uint8 [.., H, W, 3] RGB -> float [B, 3, H, W] in [0, 1]: what the LQGT dataset hands to feed_data
    (data/LQGT_dataset.py: BGR->RGB, HWC->CHW, /255)."""
    a = np.asarray(a)
    if a.ndim == 3:
        a = a[None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2))).float() / 255.


def pack_out(out, key, x, stride=3):
    """Large outputs are stored as a stride-3 subsample (exact values; 3 is coprime to every tile size, so all tile phases are hit) + a float64 digest of the whole tensor
    (n, sum, sum of squares, seeded random projection): tests/util.py::check_packed."""
    x = np.asarray(np_(x) if torch.is_tensor(x) else x)
    out[key + "_sub"] = np.ascontiguousarray(x[..., 1::stride, 2::stride])
    f = x.astype(np.float64).reshape(-1)
    r = np.random.RandomState(12345).standard_normal(f.size)
    out[key + "_dig"] = np.array([f.size, f.sum(), (f * f).sum(), (f * r).sum()], dtype=np.float64)
    out[key + "_shape"] = np.array(x.shape, dtype=np.int64)


def seeded_eps(cfg, B, h, w, tau, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) * tau for s in eps_shapes(cfg, B, h, w)]


def gen_real_net_fixture(name, preset_name, ref_sr, ref_rs, lr, hr, seed, taus=(0.0, 0.8), fit_actnorm=True, images=""):
    """Full-depth shipped configuration on real images (or a ragged multi-tile random LR): the seeded recipe, optionally with
    every ActNorm re-fitted by the REFERENCE's own data-dependent initialisation on this very HR / LR pair (ActNorms.py:29-43;
    the activations are then unit-variance per channel, the regime of a trained net), then inverse passes with seeded eps
    (replayed into the reference) and the NLL. eps / noise are regenerated from their seeds on the GPU box."""
    cfg = preset(preset_name)
    net, params = build(ref_sr if cfg.sr else ref_rs, cfg, seed)
    B, _, h, w = lr.shape
    out = {"preset": preset_name, "seed": seed, "images": images, "B": B, "h": h, "w": w}
    if not images:          # seeded random inputs: regenerated on the GPU box (tests/util.py::real_inputs)
        gi = torch.Generator().manual_seed(seed + 500)
        lr = torch.rand(lr.shape, generator=gi)
        hr = torch.rand(hr.shape, generator=gi)
        out["input_seed"] = seed + 500
    dg = param_digest(params)
    out["digest"] = np.array([dg["n"], dg["sum"], dg["sumsq"], dg["probe"]], dtype=np.float64)
    an = [(k, m) for k, m in net.named_modules() if "ActNorm" in type(m).__name__]
    noise_seed = seed + 1000
    with torch.no_grad():
        if fit_actnorm:
            for _, m in an:
                m.bias.data.zero_()
                m.logs.data.zero_()
                m.inited = False
            net.train()
            noise0 = torch.rand(hr.shape, generator=torch.Generator().manual_seed(noise_seed))
            with Capture(replay_rand=[noise0]):
                if cfg.sr:
                    net(hr=hr, lr=lr, reverse=False)
                else:
                    net(hr=hr, reverse=False)
            assert all(m.inited for _, m in an)
            net.eval()
            out["an_keys"] = np.array([k for k, _ in an])
            for i, (k, m) in enumerate(an):
                out["an_bias_%d" % i] = np_(m.bias).reshape(-1)
                out["an_logs_%d" % i] = np_(m.logs).reshape(-1)
            print("  %s: %d ActNorms fitted by the reference, logs range [%.3f, %.3f]" % (
                name, len(an), min(float(m.logs.min()) for _, m in an), max(float(m.logs.max()) for _, m in an)))
        out["fit_actnorm"] = bool(fit_actnorm)
        out["noise_seed"] = noise_seed
        for ti, tau in enumerate(taus):
            es = seed + 2000 + ti
            eps = seeded_eps(cfg, B, h, w, tau, es)
            with Capture(replay_normal=[e.clone() for e in eps]) as cap:
                y_raw = net.flow(z=lr, eps_std=tau, reverse=True)
            if tau > 0:
                assert len(cap.normal) == len(eps)
            out["inv%d_tau" % ti] = np.float64(tau)
            out["inv%d_eps_seed" % ti] = es
            pack_out(out, "inv%d_raw" % ti, y_raw)
            frac = float(((y_raw < 0) | (y_raw > 1)).float().mean())
            print("  %s tau=%.1f raw range [%.3f, %.3f] clamped frac %.3f" % (name, tau, float(y_raw.min()), float(y_raw.max()), frac))
            assert torch.isfinite(y_raw).all()
        noise = torch.rand(hr.shape, generator=torch.Generator().manual_seed(noise_seed))
        if cfg.sr:
            with Capture(replay_rand=[noise]):
                lr_hat, nll = net(hr=hr, lr=lr, reverse=False)
            with Capture(replay_rand=[noise]):
                _, nll_self = net(hr=hr, lr=lr_hat, reverse=False)
            out.update(fwd_nll=np.float64(float(nll)), fwd_nll_self=np.float64(float(nll_self)), fwd_lr=np_(lr_hat))
            print("  %s nll %.6f nll_self %.6f" % (name, float(nll), float(nll_self)))
        else:
            lr_hat, z1, z2 = net(hr=hr, reverse=False)
            out["fwd_lr"] = np_(lr_hat)
            pack_out(out, "fwd_z1", z1)
            pack_out(out, "fwd_z2", z2)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def gen_checkpoint_fixture(name, preset_name, ref_sr, ref_rs, seed):
    """Checkpoint fidelity, both directions (base_model.py:79-120). (1) reference -> ours: the REFERENCE module wrapped in
    nn.DataParallel (HCFlow_SR_model.py:33-36) is initialised by the reference's own constructors (+ small seeded values for
    the zero-initialised tensors so that it computes something), its `state_dict()` -- 'module.'-prefixed keys -- is what
    torch.save would write; stored here as arrays in state-dict order together with one reference output. (2) ours ->
    reference: the state dict in the fixture IS checked to load strictly into our class (tests) and `build()` already loads our
    tensors into the reference strictly."""
    cfg = preset(preset_name)
    ref_cls = ref_sr if cfg.sr else ref_rs
    torch.manual_seed(seed)
    np.random.seed(seed)
    net = ref_cls(opt=cfg.to_opt(), step=0)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in net.state_dict().items():
            if "haar_weights" in k:
                continue
            if float(v.abs().max()) == 0.0:                      # ActNorm bias / logs, Conv2dZeros: zero in a fresh reference net
                v.copy_(torch.randn(v.shape, generator=g) * 0.02)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net.eval()
    wrapped = torch.nn.DataParallel(net)
    sd = wrapped.state_dict()
    keys = list(sd.keys())
    assert all(k.startswith("module.") for k in keys)
    assert [k[7:] for k in keys] == [k for k, _, _ in param_spec(cfg)]
    out = {"preset": preset_name, "keys": np.array(keys)}
    for i, k in enumerate(keys):
        out["t_%d" % i] = np_(sd[k])
    B, h, w = 2, 10, 12
    lr = torch.rand(B, 3, h, w, generator=g)
    hr = torch.rand(B, 3, h * cfg.scale, w * cfg.scale, generator=g)
    eps = seeded_eps(cfg, B, h, w, 0.8, seed + 5)
    with torch.no_grad():
        with Capture(replay_normal=[e.clone() for e in eps]):
            y = net(lr=lr, eps_std=0.8, reverse=True)
        out.update(lr=np_(lr), hr=np_(hr), eps_seed=seed + 5, inv_out=np_(y))
        if cfg.sr:
            noise = torch.rand(hr.shape, generator=g)
            with Capture(replay_rand=[noise]):
                lr_hat, nll = net(hr=hr, lr=lr, reverse=False)
            out.update(fwd_noise=np_(noise), fwd_nll=np.float64(float(nll)))
        else:
            lr_hat, z1, z2 = net(hr=hr, reverse=False)
            out.update(fwd_lr=np_(lr_hat))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))



def gen_gan_fixture():
    """The reference's Discriminator_VGG_160 (discriminator_vgg_arch.py:68-107) and GANLoss (loss.py:19-51) run here on CPU: the
    module is built under torch.manual_seed (its own default initialisation; our class builds the same modules in the same
    order, so the GPU box regenerates identical parameters -- `param_digest` checks it), one train()-mode forward / backward
    of the discriminator step of HCFlow_SR_model.optimize_parameters (:258-285, gan_type 'gan') on seeded inputs; stored:
    key / shape table, outputs, losses, per-parameter gradient digests, the BatchNorm running statistics after the step."""
    import types
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tv.models)
    import importlib.util as ilu
    spec = ilu.spec_from_file_location("ref_discriminator_vgg_arch", os.path.join(REF, "models", "modules", "discriminator_vgg_arch.py"))
    D = ilu.module_from_spec(spec)
    spec.loader.exec_module(D)
    spec2 = ilu.spec_from_file_location("ref_loss", os.path.join(REF, "models", "modules", "loss.py"))
    Lm = ilu.module_from_spec(spec2)
    spec2.loader.exec_module(Lm)
    torch.manual_seed(123)
    net = D.Discriminator_VGG_160(3, 64)
    net.train()
    sd = net.state_dict()
    out = {"keys": np.array(list(sd.keys())), "shapes": np.array([",".join(str(v) for v in t_.shape) for t_ in sd.values()]),
           "seed": 123}
    out["param_digest"] = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    g = torch.Generator().manual_seed(7)
    real = torch.rand(2, 3, 160, 160, generator=g)
    fake = torch.rand(2, 3, 160, 160, generator=g)
    cri = Lm.GANLoss("gan", 1.0, 0.0)
    pred_real = net(real)
    pred_fake = net(fake)
    l_real, l_fake = cri(pred_real, True), cri(pred_fake, False)
    (l_real + l_fake).backward()
    out.update(input_seed=7, pred_real=np_(pred_real), pred_fake=np_(pred_fake), l_real=np.float64(float(l_real)),
               l_fake=np.float64(float(l_fake)))
    out["grad_digest"] = np.array([grad_digest(np_(p.grad), i) for i, (k, p) in enumerate(net.named_parameters())])
    out["grad_keys"] = np.array([k for k, _ in net.named_parameters()])
    for k, v in net.state_dict().items():
        if "running_" in k and k.startswith("bn4_1"):
            out["after_" + k] = np_(v)
    # GANLoss table: every type on a fixed logit vector
    x = torch.linspace(-2, 2, 7).view(7, 1)
    for t_ in ("gan", "ragan", "lsgan", "wgan-gp"):
        c = Lm.GANLoss(t_, 1.0, 0.0)
        out["ganloss_" + t_] = np.array([float(c(x, True)), float(c(x, False))])
    path = os.path.join(HERE, "gan_discriminator.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), "pred_real", np_(pred_real).ravel(), "losses", float(l_real), float(l_fake))


def main():
    torch.set_num_threads(8)
    ref_sr, ref_rs = import_reference()
    only = sys.argv[1] if len(sys.argv) > 1 else "all"
    if only in ("all", "real"):
        gen_real_images_fixture()
        im = np.load(os.path.join(HERE, "real_images.npz"))
        # full depth, ragged multi-tile LR (24 x 72 -> 3 x 3 conv tiles of 8 x 32 at level 1, ragged last tile)
        gen_real_net_fixture("net_sr4_full_ragged", "SR_DF2K_4X", ref_sr, ref_rs, torch.empty(1, 3, 24, 72),
                             torch.empty(1, 3, 96, 288), seed=71, fit_actnorm=False)
        gen_real_net_fixture("net_rescale_full_ragged", "Rescaling_DF2K_4X", ref_sr, ref_rs, torch.empty(1, 3, 24, 72),
                             torch.empty(1, 3, 96, 288), seed=72, taus=(0.0, 1.0), fit_actnorm=False)
        # the reference's own example images, ActNorms fitted by the reference's data-dependent init on them
        gen_real_net_fixture("net_sr4_real", "SR_DF2K_4X", ref_sr, ref_rs, img_tensor(im["butterfly_lr"]),
                             img_tensor(im["butterfly_hr"]), seed=73, images="butterfly")
        gen_real_net_fixture("net_sr8_real", "SR_CelebA_8X", ref_sr, ref_rs, img_tensor(im["face_lr"]),
                             img_tensor(im["face_hr"]), seed=74, images="face")
        gen_real_net_fixture("net_rescale_real", "Rescaling_DF2K_4X", ref_sr, ref_rs, img_tensor(im["butterfly_lr"]),
                             img_tensor(im["butterfly_hr"]), seed=75, taus=(0.0, 1.0), images="butterfly")
        if only == "real":
            return
    if only in ("all", "var"):
        # depth / split / trunk variants of every net family (the option space the fuzz tests draw from, tests/test_gpu_fuzz.py):
        # the oracle shares hcflow_amd.config.layer_plan with the product, so each family also gets fixtures from the REFERENCE's
        # own FlowNet constructors -- incl. the corner cases K = 1, after = 0, after = K, an empty first trunk
        gen_net_fixture("net_var_sr4_a", "SR_4X_tiny@K=1,3,2;after=0,3;nb=0,1", ref_sr, ref_rs, B=2, h=6, w=10, seed=101)
        gen_net_fixture("net_var_sr4_b", "SR_4X_tiny@K=3,2,2;after=3,1;nb=2,2", ref_sr, ref_rs, B=2, h=8, w=6, seed=102)
        gen_net_fixture("net_var_sr8_a", "SR_8X_tiny@K=2,1,3,2;after=1,0,3;nb=0,2", ref_sr, ref_rs, B=2, h=4, w=6, seed=103)
        gen_net_fixture("net_var_sr8_b", "SR_8X_tiny@K=1,4,2,2;after=1,2,0;nb=1,1", ref_sr, ref_rs, B=2, h=6, w=4, seed=104)
        gen_net_fixture("net_var_rescale_a", "Rescaling_4X_tiny@K=2,4,2;after=0,4;nb=0,1", ref_sr, ref_rs, B=2, h=6, w=10, seed=105,
                        taus=(0.0, 1.0))
        gen_net_fixture("net_var_rescale_b", "Rescaling_4X_tiny@K=3,1,2;after=2,0;nb=2,2", ref_sr, ref_rs, B=2, h=8, w=6, seed=106,
                        taus=(0.0, 1.0))
        if only == "var":
            return
    if only in ("all", "lu"):
        # LU-decomposed invertible 1x1 convs (Permutations.py:41-57,78-92) in every flow step: inverse / NLL / rescaling fixtures,
        # the NLL-step gradients of l / log_s / u, and the reverse-path gradients
        gen_net_fixture("net_sr4_tiny_lu", "SR_4X_tiny_LU", ref_sr, ref_rs, B=2, h=10, w=12, seed=91)
        gen_net_fixture("net_sr8_tiny_lu", "SR_8X_tiny_LU", ref_sr, ref_rs, B=2, h=5, w=6, seed=92)
        gen_net_fixture("net_rescale_tiny_lu", "Rescaling_4X_tiny_LU", ref_sr, ref_rs, B=2, h=10, w=12, seed=93, taus=(0.0, 1.0))
        gen_grad_fixture("grad_sr4_tiny_lu", "SR_4X_tiny_LU", ref_sr, B=2, h=10, w=12, seed=94)
        gen_rgrad_fixture("rgrad_sr4_tiny_lu", "SR_4X_tiny_LU", ref_sr, B=2, h=10, w=12, seed=95, tau=0.7)
        gen_rescale_grad_fixture("grad_rescale_tiny_lu", "Rescaling_4X_tiny_LU", ref_rs, B=2, h=10, w=12, seed=96)
        if only == "lu":
            return
    if only in ("all", "gan"):
        gen_gan_fixture()
        if only == "gan":
            return
    if only in ("all", "ckpt"):
        gen_checkpoint_fixture("ckpt_sr4_micro", "SR_4X_micro", ref_sr, ref_rs, seed=81)
        gen_checkpoint_fixture("ckpt_rescale_micro", "Rescaling_4X_micro", ref_sr, ref_rs, seed=82)
        if only == "ckpt":
            return
    if only in ("all", "metrics"):
        gen_metrics_fixture()
        if only == "metrics":
            return
    if only in ("all", "aninit"):
        gen_aninit_fixture("aninit_sr4_tiny", "SR_4X_tiny", ref_sr, ref_rs, B=2, h=10, w=12, seed=31)
        gen_aninit_fixture("aninit_sr8_tiny", "SR_8X_tiny", ref_sr, ref_rs, B=2, h=5, w=6, seed=32)
        gen_aninit_fixture("aninit_rescale_tiny", "Rescaling_4X_tiny", ref_sr, ref_rs, B=2, h=10, w=12, seed=33)
        if only == "aninit":
            return
    if only in ("all", "grad"):
        gen_grad_fixture("grad_sr4_tiny", "SR_4X_tiny", ref_sr, B=2, h=10, w=12, seed=41)
        gen_grad_fixture("grad_sr8_tiny", "SR_8X_tiny", ref_sr, B=2, h=5, w=6, seed=42)
        if only == "grad":
            return
    if only in ("all", "rgrad"):
        gen_rgrad_fixture("rgrad_sr4_tiny", "SR_4X_tiny", ref_sr, B=2, h=10, w=12, seed=51, tau=0.7)
        gen_rgrad_fixture("rgrad_sr8_tiny", "SR_8X_tiny", ref_sr, B=2, h=5, w=6, seed=52, tau=0.0)
        if only == "rgrad":
            return
    if only in ("all", "rescale_grad"):
        gen_rescale_grad_fixture("grad_rescale_tiny", "Rescaling_4X_tiny", ref_rs, B=2, h=10, w=12, seed=61)
        if only == "rescale_grad":
            return
    gen_op_fixture(ref_sr, ref_rs)
    # reduced-depth nets with the real channel widths, odd-ish spatial sizes, B=2
    gen_net_fixture("net_sr4_tiny", "SR_4X_tiny", ref_sr, ref_rs, B=2, h=10, w=12, seed=11)
    gen_net_fixture("net_sr8_tiny", "SR_8X_tiny", ref_sr, ref_rs, B=2, h=5, w=6, seed=12)
    gen_net_fixture("net_rescale_tiny", "Rescaling_4X_tiny", ref_sr, ref_rs, B=2, h=10, w=12, seed=13,
                    taus=(0.0, 1.0))
    # full-depth shipped configs on a small patch, B=1
    gen_net_fixture("net_sr4_full", "SR_DF2K_4X", ref_sr, ref_rs, B=1, h=8, w=8, seed=21)
    gen_net_fixture("net_sr8_full", "SR_CelebA_8X", ref_sr, ref_rs, B=1, h=4, w=4, seed=22)
    gen_net_fixture("net_rescale_full", "Rescaling_DF2K_4X", ref_sr, ref_rs, B=1, h=8, w=8, seed=23,
                    taus=(0.0, 1.0))


if __name__ == "__main__":
    main()
