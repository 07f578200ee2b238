"""The metrics oracle (oracle/metrics_oracle.py) against the reference's own functions (tests/golden/metrics.npz:
imresize, calculate_psnr, bgr2ycbcr, and -- round 6 -- ssim / calculate_ssim / calculate_psnr_ssim). OpenCV is absent here; the
reference's SSIM functions were run by tests/golden/make_golden.py over a stand-in `cv2` module that implements only
getGaussianKernel and filter2D (BORDER_REFLECT_101) from OpenCV's documentation, so constants, crop, maps, means, channel / Y
handling and crop_border are the reference's code; the fixture also records that the border rule cannot enter
(`ssim_border_dependence_*` = 0: only filter2D(...)[5:-5, 5:-5] is used). A brute-force loop evaluation and a closed-form case
check the two stand-in primitives' restatement independently."""
import math

import numpy as np
import pytest

from oracle import metrics_oracle as M
from tests.util import load_golden


@pytest.mark.parametrize("tag", ["a", "b"])
def test_psnr_y_and_imresize_match_reference(tag):
    g = load_golden("metrics")
    gt, sr = g["gt_" + tag], g["sr_" + tag]
    g8, s8 = M.tensor2img(gt) / 255.0, M.tensor2img(sr) / 255.0
    assert abs(M.calculate_psnr(g8 * 255, s8 * 255) - float(g["psnr_" + tag])) <= 1e-9
    assert np.abs(M.bgr2y(g8) - g["y_" + tag]).max() <= 1e-12
    assert abs(M.calculate_psnr(M.bgr2y(g8) * 255, M.bgr2y(s8) * 255) - float(g["psnr_y_" + tag])) <= 1e-9
    assert np.abs(M.imresize(g8, 0.25) - g["down4_" + tag]).max() <= 1e-12
    assert np.abs(M.imresize(s8, 0.5) - g["down2_" + tag]).max() <= 1e-12


@pytest.mark.parametrize("tag", ["a", "b"])
def test_ssim_matches_the_references_functions(tag):
    g = load_golden("metrics")
    g8, s8 = M.tensor2img(g["gt_" + tag]) / 255.0, M.tensor2img(g["sr_" + tag]) / 255.0
    assert abs(M.calculate_ssim(g8 * 255, s8 * 255) - float(g["ssim_" + tag])) <= 1e-12
    for cb in (0, 4):
        want = g["psnr_ssim_cb%d_%s" % (cb, tag)]
        got = M.calculate_psnr_ssim(g8, s8, cb)
        assert np.abs(np.asarray(got) - want).max() <= 1e-10, (cb, got, want)
    assert float(g["ssim_border_dependence_" + tag]) == 0.0       # zero border instead of REFLECT_101: the same value


def test_ssim_sanity():
    """Identical images give 1, the kernel sums to 1 and matches OpenCV's published
    cv2.getGaussianKernel(11, 1.5) values to their printed precision."""
    k = M.gaussian_kernel()
    assert abs(k.sum() - 1.0) <= 1e-15 and abs(k[5] - 0.26601172) <= 1e-7 and abs(k[0] - 0.00102838) <= 1e-7
    a = np.random.RandomState(1).rand(40, 50) * 255
    assert abs(M.ssim(a, a) - 1.0) <= 1e-12
    assert M.ssim(a, a[::-1]) < 0.2


def _ssim_bruteforce(a, b):
    """utils/util.py:914-934 written out position by position (no scipy, no vectorisation): weights
    w_ij = g_i g_j, g_i = exp(-(i - 5)^2 / (2 * 1.5^2)) / sum (OpenCV docs, getGaussianKernel with sigma > 0, ksize > 7)."""
    g = [math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)]
    tot = sum(g)
    g = [x / tot for x in g]
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    H, W = a.shape
    vals = []
    for y in range(5, H - 5):                 # filter2D(...)[5:-5, 5:-5]: window centred on (y, x), fully inside
        for x in range(5, W - 5):
            m1 = m2 = s11 = s22 = s12 = 0.0
            for i in range(11):
                for j in range(11):
                    w = g[i] * g[j]
                    p, q = float(a[y + i - 5, x + j - 5]), float(b[y + i - 5, x + j - 5])
                    m1 += w * p; m2 += w * q; s11 += w * p * p; s22 += w * q * q; s12 += w * p * q
            s11 -= m1 * m1; s22 -= m2 * m2; s12 -= m1 * m2
            vals.append(((2 * m1 * m2 + C1) * (2 * s12 + C2)) / ((m1 * m1 + m2 * m2 + C1) * (s11 + s22 + C2)))
    return sum(vals) / len(vals)


def test_ssim_matches_bruteforce_and_closed_form():
    rs = np.random.RandomState(3)
    a = np.round(rs.rand(14, 17) * 255)
    b = np.clip(np.round(a + rs.randn(14, 17) * 20), 0, 255)
    assert abs(M.ssim(a, b) - _ssim_bruteforce(a, b)) <= 1e-12
    # one window only (11 x 11 image): the map has a single entry
    assert abs(M.ssim(a[:11, :11], b[:11, :11]) - _ssim_bruteforce(a[:11, :11], b[:11, :11])) <= 1e-12
    # closed form: b = a + d has the same local variance and covariance as a, so every map entry is
    # (2 mu (mu + d) + C1) / (mu^2 + (mu + d)^2 + C1) with mu the Gaussian-weighted local mean of a
    d = 9.0
    k = M.gaussian_kernel()
    mu = np.array([[sum(k[i] * k[j] * a[y + i - 5, x + j - 5] for i in range(11) for j in range(11))
                    for x in range(5, a.shape[1] - 5)] for y in range(5, a.shape[0] - 5)])
    C1 = (0.01 * 255) ** 2
    want = np.mean((2 * mu * (mu + d) + C1) / (mu * mu + (mu + d) ** 2 + C1))
    assert abs(M.ssim(a, a + d) - want) <= 1e-10
    # three-channel mean (calculate_ssim, :937-955)
    a3, b3 = np.stack([a, a[::-1], a.T[:14, :14].repeat(2, 1)[:, :17]], -1), np.stack([b, b[::-1], b], -1)
    assert abs(M.calculate_ssim(a3, b3) - np.mean([_ssim_bruteforce(a3[..., c], b3[..., c]) for c in range(3)])) <= 1e-12
