"""The metrics oracle (oracle/metrics_oracle.py) against the reference's own functions (tests/golden/metrics.npz:
imresize, calculate_psnr, bgr2ycbcr). SSIM is restated from OpenCV's documented behaviour and unpinned (no cv2 here)."""
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


def test_ssim_sanity():
    """Unpinned restatement: identical images give 1, the kernel sums to 1 and matches OpenCV's published
    cv2.getGaussianKernel(11, 1.5) values to their printed precision."""
    k = M.gaussian_kernel()
    assert abs(k.sum() - 1.0) <= 1e-15 and abs(k[5] - 0.26601172) <= 1e-7 and abs(k[0] - 0.00102838) <= 1e-7
    a = np.random.RandomState(1).rand(40, 50) * 255
    assert abs(M.ssim(a, a) - 1.0) <= 1e-12
    assert M.ssim(a, a[::-1]) < 0.2
