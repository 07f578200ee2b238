"""-m gpu: the auxiliary nets of the HCFlow+ / ++ recipes (hcflow_amd/gan.py: every conv through hcf_aux_conv2d /
hcf_aux_conv2d_backward) against (a) the REFERENCE's Discriminator_VGG_160 + GANLoss discriminator step
(tests/golden/gan_discriminator.npz, generated from the reference; HCFlow_SR_model.py:258-285) and (b) stock PyTorch ops on the
same parameters (forward, input gradient, every parameter gradient)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from hcflow_amd import gan
from tests.util import load_golden, maxdiff

pytestmark = pytest.mark.gpu


def _grad_digest(g, i):
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    r = np.random.RandomState(1000 + i).standard_normal(g.size)
    return float(np.sqrt((g * g).sum())), float(g.sum()), float((g * r).sum())


def _bce_gan(pred, target_is_real):
    """What the reference's own loss.GANLoss computes for gan / ragan (loss.py:26-27,44-51); the class itself is not part of this
    package (loss.py stays the reference's file)."""
    return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, 1.0 if target_is_real else 0.0))


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_discriminator_step_matches_the_reference(precision):
    g = load_golden("gan_discriminator")
    torch.manual_seed(int(g["seed"]))
    net = gan.Discriminator_VGG_160(3, 64).cuda().train().set_precision(precision)
    gi = torch.Generator().manual_seed(int(g["input_seed"]))
    real = torch.rand(2, 3, 160, 160, generator=gi).cuda()
    fake = torch.rand(2, 3, 160, 160, generator=gi).cuda()
    cri = _bce_gan            # loss.GANLoss("gan", 1.0, 0.0) of the reference (loss.py:19-51) = BCE with logits on constant labels
    pred_real, pred_fake = net(real), net(fake)
    assert maxdiff(pred_real, g["pred_real"]) <= 2e-4 and maxdiff(pred_fake, g["pred_fake"]) <= 2e-4
    l_real, l_fake = cri(pred_real, True), cri(pred_fake, False)
    assert abs(float(l_real.detach()) - float(g["l_real"])) <= 1e-4 and abs(float(l_fake.detach()) - float(g["l_fake"])) <= 1e-4
    (l_real + l_fake).backward()
    keys = [str(k) for k in g["grad_keys"]]
    assert [k for k, _ in net.named_parameters()] == keys
    for i, (k, p) in enumerate(net.named_parameters()):
        n2, s1, pr = _grad_digest(p.grad.cpu().numpy(), i)
        w2, ws, wp = [float(v) for v in g["grad_digest"][i]]
        assert abs(n2 - w2) <= 2e-3 * max(w2, 1e-8), (k, n2, w2)
        assert abs(pr - wp) <= 2e-3 * max(w2, 1e-8) * 3, (k, pr, wp)
    for k, v in net.state_dict().items():                   # BatchNorm running statistics after two train() forwards
        if "after_" + k in g.files:
            assert maxdiff(v, g["after_" + k]) <= 1e-4 * max(1.0, float(np.abs(g["after_" + k]).max())), k


def _stock_discriminator(net, x):
    """The reference's forward (discriminator_vgg_arch.py:92-105) with stock PyTorch ops on `net`'s own parameter modules."""
    lr = lambda t: F.leaky_relu(t, 0.2)
    fea = lr(F.conv2d(x, net.conv0_0.weight, net.conv0_0.bias, 1, 1))
    fea = lr(net.bn0_1(F.conv2d(fea, net.conv0_1.weight, None, 2, 1)))
    for i in range(1, 5):
        c0, b0, c1, b1 = (getattr(net, n % i) for n in ("conv%d_0", "bn%d_0", "conv%d_1", "bn%d_1"))
        fea = lr(b0(F.conv2d(fea, c0.weight, None, 1, 1)))
        fea = lr(b1(F.conv2d(fea, c1.weight, None, 2, 1)))
    fea = lr(net.linear1(fea.reshape(fea.size(0), -1)))
    return net.linear2(fea)


def test_discriminator_gradients_match_stock_pytorch_ops():
    """Generator-side use (HCFlow_SR_model.py:232-246): gradient w.r.t. the INPUT image and every parameter, eval() mode."""
    torch.manual_seed(5)
    net = gan.Discriminator_VGG_160(3, 64).cuda().eval()
    with torch.no_grad():                                    # non-trivial running statistics
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
    ref = copy.deepcopy(net).double()
    x = torch.rand(2, 3, 160, 160, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    out = net(x)
    want = _stock_discriminator(ref, xd)
    assert maxdiff(out, want) <= 1e-4 * max(1.0, float(want.abs().max()))
    out.square().sum().backward()
    want.square().sum().backward()
    assert maxdiff(x.grad, xd.grad) <= 2e-4 * float(xd.grad.abs().max())
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert maxdiff(p.grad, q.grad) <= 2e-4 * max(float(q.grad.abs().max()), 1e-12), k


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_vgg_features_match_stock_pytorch_ops(precision):
    """VGGFeatureExtractor (discriminator_vgg_arch.py:110-137) on random weights: features and the gradient that reaches fake_H."""
    torch.manual_seed(9)
    net = gan.VGGFeatureExtractor(feature_layer=34, use_bn=False, use_input_norm=True, device=torch.device("cuda")).cuda().eval()
    net.set_precision(precision)
    with torch.no_grad():        # variance-preserving weights: the default init shrinks the activations to ~1e-8 over 16 layers,
        for m in net.features:   # far below anything a (pretrained) VGG produces and below the f16 split's absolute floor
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu")
                m.bias.normal_(0, 0.05)
    ref = copy.deepcopy(net).double()
    x = torch.rand(2, 3, 64, 96, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    fea = net(x)
    want = ref.features((xd - ref.mean) / ref.std)
    assert fea.shape == want.shape == (2, 512, 4, 6)
    assert maxdiff(fea, want) <= 1e-4 * max(1.0, float(want.abs().max()))
    # gradient reaching fake_H through the feature loss (HCFlow_SR_model.py:60-66, 226-231). ReLU units whose pre-activation sits
    # within rounding of zero switch on / off between an fp32 and an fp64 evaluation, so the comparison is in the L2 norm
    # (a handful of such units move single pixels by ~1 %), with a loose bound on the worst pixel
    tgt = torch.randn(fea.shape, generator=torch.Generator().manual_seed(1)).cuda()
    F.mse_loss(fea, tgt).backward()
    F.mse_loss(want, tgt.double()).backward()
    err = (x.grad.double() - xd.grad)
    # measured 3e-7 (no unit flipped) ... 2e-2 (tools/dbg_vgg_grad.py: it jumps with the truncation depth while every single conv's
    # dx / dw / db is at 1e-7, test_aux_conv_op_matches_torch): ONE flipped unit out of ~1.5 M active ones moves the L2 norm by ~1e-3
    assert float(err.norm()) <= 5e-2 * float(xd.grad.norm()), (float(err.norm()), float(xd.grad.norm()))


@pytest.mark.parametrize("cin,cout,H,W,act", [(64, 64, 64, 96, 0), (64, 64, 64, 96, 1), (128, 128, 32, 48, 2), (3, 64, 64, 96, 1),
                                               (512, 512, 8, 12, 1), (64, 128, 33, 47, 1), (256, 64, 16, 24, 0)])
def test_aux_conv_op_matches_torch(cin, cout, H, W, act):
    """hcf_aux_conv2d / hcf_aux_conv2d_backward on one layer: output, dL/dx, dL/dw, dL/db against an fp64 torch evaluation, and
    bit-reproducible from run to run (fixed-order weight gradient)."""
    torch.manual_seed(0)
    work = {}
    x = torch.randn(2, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda") * 0.05
    res = []
    for rep in range(2):
        xn = gan._nhwc(x).requires_grad_(True)
        wn, bn = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = gan._ConvNHWC.apply(xn, wn, bn, act, 0, work, [])
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda()
        (y * gy).sum().backward()
        res.append((y.detach().clone(), xn.grad.clone(), wn.grad.clone(), bn.grad.clone()))
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, bd, 1, 1)
    yd = F.relu(yd) if act == 1 else F.leaky_relu(yd, 0.2) if act == 2 else yd
    (yd * gy[..., :cout].permute(0, 3, 1, 2).double()).sum().backward()
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    y0, gx0, gw0, gb0 = res[0]
    assert rel(y0[..., :cout].permute(0, 3, 1, 2), yd.detach()) <= 5e-6
    assert rel(gx0[..., :cin].permute(0, 3, 1, 2), xd.grad) <= 5e-6
    assert rel(gw0, wd.grad) <= 5e-6 and rel(gb0, bd.grad) <= 5e-6
    assert all(torch.equal(a_, b_) for a_, b_ in zip(res[0], res[1]))


def test_hcflow_plus_plus_step_runs_end_to_end_on_the_engine_and_the_aux_nets():
    """One generator + discriminator step of the HCFlow++ recipe (HCFlow_SR_model.optimize_parameters :207-285 with pixel,
    feature and GAN losses, gan_type 'ragan' as in train_SR_DF2K_4X_HCFlow++.yml): netG's differentiable sampling pass, our VGG
    features, our discriminator -- every gradient finite, generator and discriminator parameters move."""
    from hcflow_amd import HCFlowNet_SR, preset, make_params
    cfg = preset("SR_4X_tiny")
    netG = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    netG.load_state_dict(make_params(cfg, 11), strict=True)
    for m in netG.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    netG = netG.cuda().train()
    torch.manual_seed(3)
    netD = gan.Discriminator_VGG_160(3, 64).cuda().train()
    netF = gan.VGGFeatureExtractor(feature_layer=34, use_bn=False, use_input_norm=True, device=torch.device("cuda")).cuda().eval()
    cri_gan = _bce_gan        # the reference's GANLoss("ragan", 1.0, 0.0): the same criterion on relativistic logits
    optG = torch.optim.Adam([p for p in netG.parameters() if p.requires_grad], lr=1e-4)
    optD = torch.optim.Adam(netD.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    lr = torch.rand(2, 3, 40, 40, generator=g).cuda()
    real = torch.rand(2, 3, 160, 160, generator=g).cuda()
    w0 = netG.flow.level0_condFlow.conv_first.weight.detach().clone()
    d0 = netD.conv0_0.weight.detach().clone()
    # (1) G: feature + GAN losses on fake_H
    optG.zero_grad()
    fake = netG(lr=lr, z=None, u=None, eps_std=0.8, reverse=True, seed=5)
    l_fea = F.l1_loss(netF(fake), netF(real).detach())
    for p in netD.parameters():
        p.requires_grad = False
    pred_fake = netD(fake)
    pred_real = netD(real).detach()
    l_gan = 5e-3 * (cri_gan(pred_real - pred_fake.mean(), False) + cri_gan(pred_fake - pred_real.mean(), True)) / 2
    (l_fea + l_gan).backward()
    assert all(torch.isfinite(p.grad).all() for p in netG.parameters() if p.grad is not None)
    optG.step()
    # (2) D
    for p in netD.parameters():
        p.requires_grad = True
    optD.zero_grad()
    pred_real = netD(real)
    pred_fake = netD(fake.detach())
    l_d = (cri_gan(pred_real - pred_fake.mean(), True) + cri_gan(pred_fake - pred_real.mean(), False)) / 2
    l_d.backward()
    assert all(torch.isfinite(p.grad).all() for p in netD.parameters())
    optD.step()
    assert not torch.equal(netG.flow.level0_condFlow.conv_first.weight.detach(), w0)
    assert not torch.equal(netD.conv0_0.weight.detach(), d0)
    assert bool(torch.isfinite(l_d)) and bool(torch.isfinite(l_fea)) and bool(torch.isfinite(l_gan))


def test_f16x3_overflow_retry_leaves_batchnorm_statistics_of_one_exact_pass():
    """A speculative f16x3 pass whose input leaves the f16 range is thrown away and redone exactly; its train()-mode BatchNorm
    updates (on inf / NaN activations) must not survive: running statistics and num_batches_tracked equal those of ONE exact
    pass (what gets saved and used in eval())."""
    torch.manual_seed(3)
    a = gan.Discriminator_VGG_160(3, 64).cuda().train().set_precision("f16x3")
    b = copy.deepcopy(a).set_precision("exact")
    x = torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(9)).cuda() * 3e5      # |x| > 65504: not splittable
    with torch.no_grad():
        ya, yb = a(x), b(x)
    assert torch.isfinite(ya).all() and torch.equal(ya, yb)
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        if "running_" in k or "num_batches" in k:
            assert torch.isfinite(u.float()).all() and torch.equal(u, v), k
    assert int(a.bn0_1.num_batches_tracked) == 1
