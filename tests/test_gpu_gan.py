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


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_discriminator_step_matches_the_reference(precision):
    g = load_golden("gan_discriminator")
    torch.manual_seed(int(g["seed"]))
    net = gan.Discriminator_VGG_160(3, 64).cuda().train().set_precision(precision)
    gi = torch.Generator().manual_seed(int(g["input_seed"]))
    real = torch.rand(2, 3, 160, 160, generator=gi).cuda()
    fake = torch.rand(2, 3, 160, 160, generator=gi).cuda()
    cri = gan.GANLoss("gan", 1.0, 0.0)
    pred_real, pred_fake = net(real), net(fake)
    assert maxdiff(pred_real, g["pred_real"]) <= 2e-4 and maxdiff(pred_fake, g["pred_fake"]) <= 2e-4
    l_real, l_fake = cri(pred_real, True), cri(pred_fake, False)
    assert abs(float(l_real) - float(g["l_real"])) <= 1e-4 and abs(float(l_fake) - float(g["l_fake"])) <= 1e-4
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
    ref = copy.deepcopy(net).double()
    x = torch.rand(2, 3, 64, 96, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    fea = net(x)
    want = ref.features((xd - ref.mean) / ref.std)
    assert fea.shape == want.shape == (2, 512, 4, 6)
    assert maxdiff(fea, want) <= 1e-4 * max(1.0, float(want.abs().max()))
    F.l1_loss(fea, torch.zeros_like(fea)).backward()          # cri_fea = L1 (HCFlow_SR_model.py:60-66)
    F.l1_loss(want, torch.zeros_like(want)).backward()
    assert maxdiff(x.grad, xd.grad) <= 3e-4 * float(xd.grad.abs().max())
