"""Validation metrics on the GPU: the per-image block of the reference's test loop (test_HCFlow.py:103-182) --
``tensor2img`` + ``calculate_psnr_ssim`` (utils/util.py:790-816, 898-982), optionally on MATLAB-style bicubic
down-scaled copies (utils/imresize.py), and the sample diversity (``std`` over samples, test_HCFlow.py:165) --
computed by HIP kernels in float64 (include/hcflow.h: hcf_metric_*). No CPU fallback."""
import ctypes as C
import math
from typing import Dict, List

import numpy as np
import torch

from . import _lib

KEYS = ["psnr", "ssim", "psnr_y", "ssim_y", "bic_psnr", "bic_ssim", "bic_psnr_y", "bic_ssim_y"]


def _prep(t):
    assert t.is_cuda and t.dim() == 4 and t.shape[1] == 3, "expected a CUDA tensor [B,3,H,W]"
    return t.detach().to(torch.float32).contiguous()


def psnr_ssim(gt: torch.Tensor, sr: torch.Tensor, crop_border: int = 0, scale: int = 0) -> List[Dict[str, float]]:
    """Per image: PSNR / SSIM / PSNR_Y / SSIM_Y of tensor2img(gt) vs tensor2img(sr) (borders cropped), and for
    ``scale > 1`` the same four on the bicubic down-scaled pair (the "bicHR" columns of the reference's log line)."""
    lib = _lib.load()
    gt, sr = _prep(gt), _prep(sr)
    assert gt.shape == sr.shape
    B, _, H, W = gt.shape
    out = np.zeros((B, 8), dtype=np.float64)
    with torch.cuda.device(gt.device):
        rc = lib.hcf_metric_psnr_ssim(gt.data_ptr(), sr.data_ptr(), B, H, W, int(crop_border), int(scale),
                                      out.ctypes.data_as(C.c_void_p),
                                      C.c_void_p(torch.cuda.current_stream(gt.device).cuda_stream))
    _lib.check(rc, None, "hcf_metric_psnr_ssim")
    return [dict(zip(KEYS, map(float, row))) for row in out]


def imresize_down(x: torch.Tensor, scale: int) -> np.ndarray:
    """imresize(tensor2img(x) / 255., 1 / scale) * 255 as a float64 array [B, h, w, 3] in BGR order (0..255 units)."""
    lib = _lib.load()
    x = _prep(x)
    B, _, H, W = x.shape
    h, w = math.ceil(H / scale), math.ceil(W / scale)
    out = np.zeros((B, 3, h, w), dtype=np.float64)
    with torch.cuda.device(x.device):
        rc = lib.hcf_metric_imresize_down(x.data_ptr(), B, H, W, int(scale), out.ctypes.data_as(C.c_void_p),
                                          C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
    _lib.check(rc, None, "hcf_metric_imresize_down")
    return np.transpose(out, (0, 2, 3, 1))


def diversity(samples: List[torch.Tensor]) -> float:
    """torch.cat([s.unsqueeze(0) * 255 for s in samples], 0).std([0]).mean() (test_HCFlow.py:127,165)."""
    return float((torch.stack([s.float() for s in samples], 0) * 255).std(0).mean())
