"""Batch > 1 test loading and batched evaluation.

The reference's test loader is hard-wired to one image per step (``codes/data/__init__.py:24``: ``DataLoader(dataset,
batch_size=1, shuffle=False, num_workers=0)``) and its metrics loop (``test_HCFlow.py:85-182``) converts every sample to a
uint8 numpy image on the host. On an MI355X a single 160x160 LR image leaves the GPU idle most of the time (B = 1: 21.6 ms per
image, B = 16: 8.6 ms per image). This module is the drop-in for those two pieces:

* ``batched_test_loader(dataset, batch_size)`` walks ANY reference dataset (``__len__`` / ``__getitem__`` returning the
  reference's dict: ``LQ`` [3,h,w], optional ``GT`` [3,H,W], ``LQ_path`` / ``GT_path``) in order and stacks consecutive items of
  identical shape into one batch (test sets mix image sizes: a size change closes the batch), yielding the dict
  ``feed_data`` expects (``HCFlow_SR_model.py:177-182``) with lists of paths;
* ``evaluate_batch(net, batch, heats, n_sample, scale, crop_border)`` runs what ``HCFlowSRModel.test()`` runs
  (``HCFlow_SR_model.py:281-301``: NLL pass + one sampling pass per heat and sample) on the whole batch and computes the log
  line's numbers per image on the device (hcflow_amd/metrics.py: PSNR / SSIM / PSNR_Y / SSIM_Y of SR vs GT with the border
  crop, the same on the bicubic down-scaled pair, LR consistency, sample diversity). LPIPS needs the AlexNet weights the
  ``lpips`` package downloads and is not computed (``lpips`` is absent from this image).

The numbers per image are identical to feeding the images one at a time (every op of the path is per sample)."""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional, Sequence

import torch

from . import metrics as M


def batched_test_loader(dataset, batch_size: int = 16, keys: Sequence[str] = ("LQ", "GT")) -> Iterator[Dict]:
    """Order-preserving batches of same-shaped items of a reference test dataset."""
    assert batch_size >= 1
    pend: List[Dict] = []

    def shape_of(item):
        return tuple(tuple(item[k].shape) for k in keys if k in item and torch.is_tensor(item[k]))

    def flush():
        out: Dict = {}
        for k in pend[0]:
            vals = [it[k] for it in pend]
            out[k] = torch.stack(vals, 0) if torch.is_tensor(vals[0]) else vals
        pend.clear()
        return out

    for i in range(len(dataset)):
        item = dataset[i]
        if pend and (len(pend) == batch_size or shape_of(item) != shape_of(pend[0])):
            yield flush()
        pend.append(item)
    if pend:
        yield flush()


def evaluate_batch(net, batch: Dict, heats: Iterable[float], n_sample: int, scale: int, crop_border: Optional[int] = None,
                   seed: Optional[int] = None, noise: Optional[torch.Tensor] = None, lpips_fn=None) -> List[Dict]:
    """Per image of the batch: {"nll": float, "lr": {psnr, ssim, psnr_y, ssim_y} of LR^ vs LQ (SR nets), and per heat:
    {"psnr", "ssim", "psnr_y", "ssim_y", "bic_psnr", ... (means over the samples), "diversity"}} -- the numbers of the
    reference's per-image log line (test_HCFlow.py:166-175). LPIPS needs the ``lpips`` package's pretrained AlexNet: pass the
    caller's own module as ``lpips_fn`` (``lpips.LPIPS(net='alex').to('cuda')``, test_HCFlow.py:48) and every heat entry gets
    "lpips" = the mean over the samples of ``lpips_fn(2 gt - 1, 2 sr - 1)`` per image (:132-133, :168), evaluated on the whole
    batch at once; without it the key is absent. ``net`` is an eval()-mode HCFlowNet_SR on a GPU.
    ``seed`` fixes the device draws of the samples, ``noise`` ([B,3,H,W] in [0,1)) replaces the dequantisation noise of the
    NLL pass (HCFlowNet_SR_arch.py:52); both default to fresh draws, as in the reference."""
    dev = next(net.parameters()).device
    lq = batch["LQ"].to(dev)
    gt = batch["GT"].to(dev) if "GT" in batch else None
    B = lq.shape[0]
    crop = scale if crop_border is None else crop_border          # test_HCFlow.py:48
    res: List[Dict] = [dict() for _ in range(B)]
    with torch.no_grad():
        if gt is not None:
            # per-image NLL: the engine returns the objective (logdet + log p) of every sample; nll_b = -obj_b / (ln 2 * H W)
            # (HCFlowNet_SR_arch.py:62-66: the reference's nll.mean() over a batch of one is exactly this)
            lr_hat, _, obj, _ = net.normal_flow_diracLR(gt, lq, noise=noise, return_internals=True)
            pix = float(gt.shape[2] * gt.shape[3])
            for b, o in enumerate(obj.double().cpu().tolist()):
                res[b]["nll"] = -o / (0.6931471805599453 * pix)
            for b, m in enumerate(M.psnr_ssim(lq, lr_hat, 0, 0)):
                res[b]["lr"] = {k: m[k] for k in ("psnr", "ssim", "psnr_y", "ssim_y")}
        for hi, heat in enumerate(heats):
            samples = []
            acc = [dict.fromkeys(M.KEYS, 0.0) for _ in range(B)]
            lp = [0.0] * B
            for s in range(n_sample):
                kw = {} if seed is None else {"seed": seed + 1000 * hi + s}
                # every heat / sample of the sweep conditions on the SAME lq: the deepest level's conditional features (the RRDB
                # trunk: most of a Face x8 call) are computed once and reused, bit-identical outputs (arch.py: cache_cond)
                sr = net(lr=lq, z=None, u=None, eps_std=heat, reverse=True, training=False, cache_cond=True, **kw)
                samples.append(sr)
                if gt is not None:
                    for b, m in enumerate(M.psnr_ssim(gt, sr, crop, scale)):
                        for k in M.KEYS:
                            acc[b][k] += m[k] / n_sample
                    if lpips_fn is not None:
                        d = lpips_fn(2 * gt - 1, 2 * sr - 1).reshape(B, -1).mean(dim=1)      # [B, 1, 1, 1] -> one value per image
                        for b, v in enumerate(d.double().cpu().tolist()):
                            lp[b] += v / n_sample
            for b in range(B):
                ent = dict(acc[b]) if gt is not None else {}
                if gt is not None and lpips_fn is not None:
                    ent["lpips"] = lp[b]
                ent["diversity"] = M.diversity([s[b:b + 1] for s in samples]) if n_sample > 1 else 0.0
                res[b][float(heat)] = ent
    return res
