"""hcflow_amd/gan.py on the CPU side: the algebra that lets the discriminator's 4x4 stride-2 convs run on the 3x3 kernels, the
state-dict table against the reference's Discriminator_VGG_160 (fixture generated from the reference,
tests/golden/make_golden.py::gen_gan_fixture), loud failure without a GPU."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hcflow_amd import gan
from hcflow_amd._lib import HcfError
from tests.util import load_golden


def test_4x4_stride2_conv_equals_squeeze_plus_reindexed_3x3():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 12, 16, generator=g, dtype=torch.float64)
    w = torch.randn(7, 5, 4, 4, generator=g, dtype=torch.float64, requires_grad=True)
    want = F.conv2d(x, w, stride=2, padding=1)
    xs = gan.squeeze2d_nhwc(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    got = F.conv2d(xs, gan._w4s2_as_3x3(w), stride=1, padding=1)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-12
    gw_want, = torch.autograd.grad(want.square().sum(), w)
    gw_got, = torch.autograd.grad(got.square().sum(), w)
    assert float((gw_got - gw_want).abs().max()) <= 1e-9 * float(gw_want.abs().max())


def test_discriminator_state_dict_table_and_seeded_init_equal_the_reference():
    g = load_golden("gan_discriminator")
    torch.manual_seed(int(g["seed"]))
    net = gan.Discriminator_VGG_160(3, 64)
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(str(v) for v in t.shape) for t in sd.values()] == [str(s) for s in g["shapes"]]
    dig = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    assert np.allclose(dig, g["param_digest"], rtol=1e-9, atol=1e-12)      # same modules, same order -> same default init


def test_vgg_feature_extractor_layer_table():
    net = gan.VGGFeatureExtractor(feature_layer=34, use_bn=False)
    keys = list(net.state_dict().keys())
    convs = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]           # torchvision vgg19().features conv indices
    assert [k for k in keys if k.startswith("features")] == [f"features.{i}.{p}" for i in convs for p in ("weight", "bias")]
    assert len(net.features) == 35 and isinstance(net.features[34], torch.nn.Conv2d)    # conv5_4, before its ReLU
    assert not any(p.requires_grad for p in net.features.parameters())


def test_no_cpu_fallback():
    net = gan.Discriminator_VGG_160(3, 64)
    if not torch.cuda.is_available():
        with pytest.raises(HcfError):
            net(torch.rand(1, 3, 160, 160))
