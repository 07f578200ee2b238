"""-m gpu: the device metrics (hcflow_amd/metrics.py) against the metrics oracle and the reference-generated fixture."""
import numpy as np
import pytest
import torch

from oracle import metrics_oracle as M
from tests.util import load_golden

pytestmark = pytest.mark.gpu


def _pair(g, tag):
    return torch.from_numpy(g["gt_" + tag]).unsqueeze(0), torch.from_numpy(g["sr_" + tag]).unsqueeze(0)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("crop", [0, 4])
def test_psnr_ssim_matches_reference_and_oracle(tag, crop):
    from hcflow_amd import metrics
    g = load_golden("metrics")
    gt, sr = _pair(g, tag)
    r = metrics.psnr_ssim(gt.cuda(), sr.cuda(), crop_border=crop, scale=4)[0]
    g8, s8 = M.tensor2img(gt.numpy()) / 255.0, M.tensor2img(sr.numpy()) / 255.0
    want = M.calculate_psnr_ssim(g8, s8, crop)
    for k, w in zip(["psnr", "ssim", "psnr_y", "ssim_y"], want):
        assert abs(r[k] - w) <= 1e-9 * max(1.0, abs(w)), (k, r[k], w)
    if crop == 0:                                           # pinned by the reference's own functions
        assert abs(r["psnr"] - float(g["psnr_" + tag])) <= 1e-9
        assert abs(r["psnr_y"] - float(g["psnr_y_" + tag])) <= 1e-9
    # all four by the reference's calculate_psnr_ssim (utils/util.py:958-982), run over the documented cv2 stand-in of make_golden.py
    ref4 = g["psnr_ssim_cb%d_%s" % (crop, tag)]
    for k, w in zip(["psnr", "ssim", "psnr_y", "ssim_y"], ref4):
        assert abs(r[k] - float(w)) <= 1e-9 * max(1.0, abs(float(w))), (k, r[k], float(w))
    wb = M.calculate_psnr_ssim(M.imresize(g8, 0.25), M.imresize(s8, 0.25), 0)
    for k, w in zip(["bic_psnr", "bic_ssim", "bic_psnr_y", "bic_ssim_y"], wb):
        if np.isnan(w):                                     # 10-row image: the 11x11 window has no valid position
            assert np.isnan(r[k])
        else:
            assert abs(r[k] - w) <= 1e-8 * max(1.0, abs(w)), (k, r[k], w)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_imresize_matches_reference(tag):
    from hcflow_amd import metrics
    g = load_golden("metrics")
    gt, sr = _pair(g, tag)
    d4 = metrics.imresize_down(gt.cuda(), 4)[0] / 255.0
    d2 = metrics.imresize_down(sr.cuda(), 2)[0] / 255.0
    assert d4.shape == g["down4_" + tag].shape and d2.shape == g["down2_" + tag].shape
    assert np.abs(d4 - g["down4_" + tag]).max() <= 1e-12
    assert np.abs(d2 - g["down2_" + tag]).max() <= 1e-12


def test_batch_and_identical_images():
    from hcflow_amd import metrics
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 3, 40, 56, generator=g).cuda()
    y = (x + 0.03 * torch.randn(x.shape, generator=g).cuda()).clamp(0, 1)
    r = metrics.psnr_ssim(x, y)
    one = metrics.psnr_ssim(x[1:2], y[1:2])[0]
    assert all(abs(r[1][k] - one[k]) <= 1e-12 for k in one)
    same = metrics.psnr_ssim(x, x)[0]
    assert same["psnr"] == float("inf") and abs(same["ssim"] - 1.0) <= 1e-12
    assert abs(metrics.diversity([x[0], x[1], x[2]]) - float((torch.stack([x[0], x[1], x[2]]) * 255).std(0).mean())) <= 1e-4


def test_batched_evaluation_equals_one_image_at_a_time():
    """hcflow_amd/loader.py: the reference's per-image test loop (batch 1, test_HCFlow.py:85-182) run on batches -- every
    number of an image is the one it gets when fed alone (per-sample ops; tau = 0 and seeded samples)."""
    from hcflow_amd import HCFlowNet_SR, preset, make_params
    from hcflow_amd.loader import batched_test_loader, evaluate_batch
    from tests.test_loader_cpu import FakeSet
    cfg = preset("SR_4X_tiny")
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 11), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval()
    ds = FakeSet([(12, 16)] * 3 + [(16, 12)] * 2)        # LR >= 11 pixels a side: the SSIM window (util.py:907)
    heats = [0.0, 0.8]
    def noise_for(b, first):           # the dequantisation noise of dataset item i, whatever batch it lands in
        return torch.stack([torch.rand(b["GT"].shape[1:], generator=torch.Generator().manual_seed(500 + first + j))
                            for j in range(b["GT"].shape[0])], 0).cuda()
    together, alone, k = [], [], 0
    for b in batched_test_loader(ds, 4):
        together += evaluate_batch(net, b, heats, n_sample=1, scale=4, noise=noise_for(b, k))
        k += b["LQ"].shape[0]
    k = 0
    for b in batched_test_loader(ds, 1):
        alone += evaluate_batch(net, b, [0.0], n_sample=1, scale=4, noise=noise_for(b, k))
        k += 1
    assert len(together) == len(alone) == 5
    for t_, a_ in zip(together, alone):
        assert abs(t_["nll"] - a_["nll"]) <= 1e-6 * max(1.0, abs(a_["nll"]))
        for k in ("psnr", "ssim", "psnr_y", "ssim_y"):
            assert abs(t_["lr"][k] - a_["lr"][k]) <= 1e-9 * max(1.0, abs(a_["lr"][k]))
        for k in ("psnr", "ssim", "bic_psnr", "bic_ssim_y"):
            assert abs(t_[0.0][k] - a_[0.0][k]) <= 1e-9 * max(1.0, abs(a_[0.0][k])), k
        assert set(t_[0.8].keys()) >= {"psnr", "ssim", "diversity"}


def test_lpips_hook_receives_the_reference_inputs_per_image():
    """evaluate_batch(lpips_fn=...): the caller's own lpips module (the package's pretrained AlexNet cannot be rebuilt offline)
    gets (2 gt - 1, 2 sr - 1) as in test_HCFlow.py:132 and returns [B, 1, 1, 1]; the entry is the per-image mean over samples.
    Checked with a stand-in distance whose per-image value is known."""
    from hcflow_amd import HCFlowNet_SR, preset, make_params
    from hcflow_amd.loader import batched_test_loader, evaluate_batch
    from tests.test_loader_cpu import FakeSet
    cfg = preset("SR_4X_tiny")
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(make_params(cfg, 11), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.cuda().eval()
    seen = []

    def fake_lpips(a, b):              # a "distance" in the [-1, 1] domain: mean |a - b| per image, shaped like lpips' output
        seen.append((float(a.min()), float(a.max()), tuple(a.shape), tuple(b.shape)))
        return (a - b).abs().mean(dim=(1, 2, 3), keepdim=True)

    ds = FakeSet([(12, 16)] * 3)
    b = next(iter(batched_test_loader(ds, 3)))
    res = evaluate_batch(net, b, [0.0], n_sample=2, scale=4, seed=3, lpips_fn=fake_lpips)
    assert len(seen) == 2 and all(s[2] == s[3] == (3, 3, 48, 64) and -1.0 <= s[0] and s[1] <= 1.0 for s in seen)
    with torch.no_grad():
        sr = net(lr=b["LQ"].cuda(), eps_std=0.0, reverse=True)
    want = (2 * (b["GT"].cuda() - sr)).abs().mean(dim=(1, 2, 3)).cpu().tolist()           # tau = 0: both samples are identical
    for r, w in zip(res, want):
        assert abs(r[0.0]["lpips"] - w) <= 1e-6
    assert "lpips" not in evaluate_batch(net, b, [0.0], n_sample=1, scale=4)[0][0.0]
