"""Debug aid: input-gradient error of hcflow_amd.gan.VGGFeatureExtractor against an fp64 stock-PyTorch evaluation as a function of
the truncation depth (GPU box). The error jumps between 3e-7 and 2e-2 with the depth while every single conv is at 1e-7
(tools/dbg_aux_conv.py): ReLU units within rounding of zero switch between the fp32 and the fp64 evaluation."""
import copy, torch, torch.nn as nn, torch.nn.functional as F, sys
sys.path.insert(0, '.')
from hcflow_amd import gan
for fl in (0, 2, 4, 5, 7, 9, 10, 16, 18, 19, 27, 34):
    torch.manual_seed(9)
    net = gan.VGGFeatureExtractor(feature_layer=fl, use_bn=False, use_input_norm=True, device=torch.device("cuda")).cuda().eval()
    with torch.no_grad():
        for m in net.features:
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="relu"); m.bias.normal_(0, 0.05)
    ref = copy.deepcopy(net).double()
    x = torch.rand(2, 3, 64, 96, device="cuda", requires_grad=True)
    xd = x.detach().double().requires_grad_(True)
    fea = net(x); want = ref.features((xd - ref.mean) / ref.std)
    tgt = torch.randn(fea.shape, generator=torch.Generator().manual_seed(1)).cuda()
    F.mse_loss(fea, tgt).backward(); F.mse_loss(want, tgt.double()).backward()
    err = x.grad.double() - xd.grad
    print("feature_layer %2d  last=%s  fwd rel %.2e   grad relL2 %.3e  relmax %.3e" % (fl, type(net.features[fl]).__name__, float((fea.double()-want).norm()/want.norm()), float(err.norm()/xd.grad.norm()), float(err.abs().max()/xd.grad.abs().max())))
