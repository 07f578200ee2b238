"""The auxiliary nets of the HCFlow+ / HCFlow++ recipes on the MI355X conv kernels.

``HCFlow_SR_model.py:75-95`` builds, beside netG, a VGG19 feature extractor (``networks.define_F`` -> ``VGGFeatureExtractor``,
``discriminator_vgg_arch.py:110-137``) for the perceptual loss and a ``Discriminator_VGG_160`` (``:68-107``) trained with the
reference's own ``GANLoss`` (``loss.py:19-51``: stock PyTorch criteria, no kernel behind it -- it stays the reference's file and is
not restated here); ``optimize_parameters`` (``HCFlow_SR_model.py:219-285``) runs them forward and backward every
step on ``fake_H`` / ``real_H`` batches. The classes here keep the reference's constructor signatures, ``state_dict`` keys /
shapes (the parameter holders ARE ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` modules, so checkpoints of the reference
load strictly, ``load_network(..., netD)``) and call surface, and run EVERY CONVOLUTION -- > 99 % of their FLOPs -- through the
flow's own kernels on device tensors (``hcf_aux_conv2d`` / ``hcf_aux_conv2d_backward`` in ``include/hcflow.h``: fp32-MFMA or
f16x3 / Winograd forward, fp32-MFMA data gradient, fixed-order weight gradient):

* activations travel as NHWC fp32 (the engine's layout), 3-channel inputs padded to 4;
* the discriminator's 4x4 stride-2 convs are a ``squeeze2d`` (space-to-depth) followed by a 3x3 conv on 4C channels whose
  weight is the 4x4 kernel re-indexed (``_w4s2_as_3x3``; exact, differentiable);
* bias + LeakyReLU / ReLU are fused into the conv epilogue where no BatchNorm sits in between;
* BatchNorm (batch statistics in train(), running statistics in eval()), the two Linear layers, max-pooling and the losses
  are a few elementwise / reduction ops per layer and stay on stock PyTorch ops.

``VGGFeatureExtractor`` needs torchvision's pretrained VGG19 weights, which cannot be downloaded here: the layer stack is
rebuilt from the VGG19 configuration with the same ``features.N`` keys, so a torchvision ``vgg19().features`` state dict loads
strictly; without one the weights are random. No CPU fallback: ``forward`` raises ``HcfError`` off-GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

_PREC = {"exact": 0, "f16x3": 1}


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW -> dense NHWC fp32 with the channel count padded to a multiple of 4."""
    B, Cc, H, W = x.shape
    y = x.permute(0, 2, 3, 1).to(torch.float32)
    if Cc % 4:
        y = F.pad(y, (0, 4 - Cc % 4))
    return y.contiguous()


def _nchw(y: torch.Tensor, Cc: int) -> torch.Tensor:
    return y[..., :Cc].permute(0, 3, 1, 2)


def squeeze2d_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] -> [B,H/2,W/2,4C], channel order c*4 + a*2 + b (Basic.squeeze2d, Basic.py:127-141)."""
    B, H, W, Cc = x.shape
    return x.view(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, 4 * Cc).contiguous()


def _w4s2_as_3x3(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(C, O, 4, 2, 1) weight [O,C,4,4] -> the 3x3 stride-1 weight [O,4C,3,3] acting on squeeze2d(x):
    out[y,x] = sum_ij w[i,j] in[2y-1+i, 2x-1+j]; input row 2y-1+i = squeezed row y+dy, sub-row a with
    (i -> dy, a) = 0 -> (-1, 1), 1 -> (0, 0), 2 -> (0, 1), 3 -> (+1, 0); same for columns."""
    O, Cc = w.shape[0], w.shape[1]
    w3 = w.new_zeros(O, Cc, 2, 2, 3, 3)
    m = ((0, -1, 1), (1, 0, 0), (2, 0, 1), (3, 1, 0))
    for i, dy, a in m:
        for j, dx, b in m:
            w3[:, :, a, b, dy + 1, dx + 1] = w[:, :, i, j]
    return w3.reshape(O, 4 * Cc, 3, 3)


class _ConvNHWC(torch.autograd.Function):
    """y = act(conv_k(x, w) + bias) on NHWC device tensors through the C ABI; act in {0 none, 1 relu, 2 lrelu 0.2}."""

    @staticmethod
    def forward(ctx, x, w, bias, act, prec, work, flag_owner):
        if not x.is_cuda:
            raise _lib.HcfError("hcflow_amd.gan runs on MI355X only (no CPU fallback): move the module and its inputs to a GPU")
        lib = _lib.load()
        B, H, W, cs = x.shape
        cout, cin, k, _ = w.shape
        assert cs % 4 == 0 and cs >= cin and x.is_contiguous() and x.dtype == torch.float32
        w = w.contiguous()
        y = torch.empty(B, H, W, (cout + 3) & ~3, device=x.device, dtype=torch.float32)
        if y.shape[3] != cout:
            y.zero_()
        need = lib.hcf_aux_conv2d_workspace(cin, cout, k, B, H, W)
        # keyed by DEVICE and STREAM too: nn.DataParallel replicas share this dict (replicate() shallow-copies __dict__), one
        # replica thread per device (HCFlow_SR_model.py:76,94 wrap netD / netF), and a workspace holds packs, the range flag and
        # weight-gradient partials of the call in flight
        key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream, cin, cout, k, B, H, W)
        wk = work.get(key)
        if wk is None or wk.numel() < need or wk.device != x.device:
            wk = torch.zeros(need, dtype=torch.uint8, device=x.device)          # [0, 256): range flag + zero page
            work[key] = wk
        flag_owner.append(wk)
        with torch.cuda.device(x.device):
            rc = lib.hcf_aux_conv2d(x.data_ptr(), cs, cin, B, H, W, w.data_ptr(), None if bias is None else bias.contiguous().data_ptr(),
                                    cout, k, act, y.data_ptr(), y.shape[3], C.c_void_p(wk.data_ptr()), wk.numel(), prec,
                                    C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(rc, None, "hcf_aux_conv2d")
        ctx.save_for_backward(x, w, y if act else None)
        ctx.meta = (cin, cout, k, act, prec, bias is not None, wk)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        cin, cout, k, act, prec, has_bias, wk = ctx.meta
        B, H, W, cs = x.shape
        g = g.contiguous()
        if act == 1:
            g = g * (y > 0)
        elif act == 2:
            g = g * torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
        need_dx = ctx.needs_input_grad[0]
        dx = torch.zeros_like(x) if need_dx else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None       # frozen weights (VGG; netD during the G step): skipped
        with torch.cuda.device(x.device):
            rc = lib.hcf_aux_conv2d_backward(x.data_ptr(), cs, cin, B, H, W, w.data_ptr(), cout, k, g.data_ptr(), g.shape[3],
                                             None if dx is None else dx.data_ptr(), cs, None if dw is None else dw.data_ptr(),
                                             C.c_void_p(wk.data_ptr()),
                                             wk.numel(), prec, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        _lib.check(rc, None, "hcf_aux_conv2d_backward")
        db = g[..., :cout].sum(dim=(0, 1, 2)) if (has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None, None, None, None


class _AuxNet(nn.Module):
    """Shared plumbing: precision policy and the per-layer workspaces."""

    def _aux_init(self):
        object.__setattr__(self, "_work", {})
        object.__setattr__(self, "_prec", ["exact"])

    def set_precision(self, mode: str):
        """"exact" (default): fp32-MFMA convs. "f16x3": fp32-equivalent split convs for the 3x3 forward passes; an input beyond
        the f16 range is detected after the pass and the pass is redone exactly (one stream sync per forward)."""
        assert mode in _PREC
        self._prec[0] = mode
        return self

    def _conv(self, x, conv: nn.Conv2d, act: int, flags, prec):
        w = conv.weight
        if conv.kernel_size == (4, 4):                       # 4x4 stride 2 pad 1 == squeeze2d + re-indexed 3x3
            x = squeeze2d_nhwc(x)
            w = _w4s2_as_3x3(w)
        return _ConvNHWC.apply(x, w, conv.bias, act, prec, self._work, flags)

    def _run(self, body, x):
        prec = _PREC[self._prec[0]]
        flags = []
        # a speculative f16x3 pass in train() mode updates every BatchNorm's running statistics; if it is thrown away (range
        # overflow: its activations were inf / NaN) the exact re-run must start from the statistics BEFORE it
        bn_state = None
        if prec == 1 and self.training:
            bn_state = [(m, m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone())
                        for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.track_running_stats and m.running_mean is not None]
        out = body(x, flags, prec)
        if prec == 1 and flags:
            hit = torch.stack([f[:4].view(torch.int32)[0] for f in {id(f): f for f in flags}.values()]).any()
            if bool(hit):                                     # an activation left the f16 range: redo the pass exactly
                for f in flags:
                    f[:4].zero_()
                for m, mean, var, cnt in (bn_state or []):
                    with torch.no_grad():
                        m.running_mean.copy_(mean)
                        m.running_var.copy_(var)
                        m.num_batches_tracked.copy_(cnt)
                out = body(x, [], 0)
        return out


class Discriminator_VGG_160(_AuxNet):
    """Drop-in for discriminator_vgg_arch.Discriminator_VGG_160 (:68-107): same modules / state_dict, convs on our kernels."""

    def __init__(self, in_nc, nf):
        super().__init__()
        self.conv0_0 = nn.Conv2d(in_nc, nf, 3, 1, 1, bias=True)
        self.conv0_1 = nn.Conv2d(nf, nf, 4, 2, 1, bias=False)
        self.bn0_1 = nn.BatchNorm2d(nf, affine=True)
        chans = [(nf, nf * 2), (nf * 2, nf * 4), (nf * 4, nf * 8), (nf * 8, nf * 8)]
        for i, (ci, co) in enumerate(chans, start=1):
            setattr(self, "conv%d_0" % i, nn.Conv2d(ci, co, 3, 1, 1, bias=False))
            setattr(self, "bn%d_0" % i, nn.BatchNorm2d(co, affine=True))
            setattr(self, "conv%d_1" % i, nn.Conv2d(co, co, 4, 2, 1, bias=False))
            setattr(self, "bn%d_1" % i, nn.BatchNorm2d(co, affine=True))
        self.linear1 = nn.Linear(512 * 5 * 5, 100)
        self.linear2 = nn.Linear(100, 1)
        self.lrelu = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self._aux_init()

    def _bn_lrelu(self, y, bn: nn.BatchNorm2d):
        Cc = bn.num_features
        v = bn(y[..., :Cc].permute(0, 3, 1, 2))               # channels-last view: no copy; batch / running statistics as nn.BatchNorm2d
        return F.leaky_relu(v, 0.2).permute(0, 2, 3, 1).contiguous()

    def _body(self, x, flags, prec):
        fea = self._conv(_nhwc(x), self.conv0_0, 2, flags, prec)                       # bias + LeakyReLU fused
        fea = self._bn_lrelu(self._conv(fea, self.conv0_1, 0, flags, prec), self.bn0_1)
        for i in range(1, 5):
            fea = self._bn_lrelu(self._conv(fea, getattr(self, "conv%d_0" % i), 0, flags, prec), getattr(self, "bn%d_0" % i))
            fea = self._bn_lrelu(self._conv(fea, getattr(self, "conv%d_1" % i), 0, flags, prec), getattr(self, "bn%d_1" % i))
        fea = fea.permute(0, 3, 1, 2).reshape(fea.size(0), -1)                         # the reference flattens NCHW
        fea = self.lrelu(self.linear1(fea))
        return self.linear2(fea)

    def forward(self, x):
        return self._run(self._body, x)

    def reset_parameters(self):
        for layer in self.children():
            if hasattr(layer, "reset_parameters"):
                layer.reset_parameters()


_VGG19 = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def _vgg19_features(use_bn):
    layers, cin = [], 3
    for v in _VGG19:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers.append(nn.Conv2d(cin, v, kernel_size=3, padding=1))
            if use_bn:
                layers.append(nn.BatchNorm2d(v))
            layers.append(nn.ReLU(inplace=True))
            cin = v
    return layers


class VGGFeatureExtractor(_AuxNet):
    """Drop-in for discriminator_vgg_arch.VGGFeatureExtractor (:110-137): VGG19 ``features[:feature_layer + 1]`` (34 = conv5_4
    before its ReLU), input normalisation, frozen weights. The stack has torchvision's ``features.N`` keys; pretrained weights
    are whatever the caller loads (none ship here)."""

    def __init__(self, feature_layer=34, use_bn=False, use_input_norm=True, device=torch.device("cpu")):
        super().__init__()
        self.use_input_norm = use_input_norm
        if self.use_input_norm:
            self.register_buffer("mean", torch.Tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1).to(device))
            self.register_buffer("std", torch.Tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1).to(device))
        self.features = nn.Sequential(*_vgg19_features(use_bn)[:(feature_layer + 1)])
        for k, v in self.features.named_parameters():
            v.requires_grad = False
        self._aux_init()

    def _body(self, x, flags, prec):
        if self.use_input_norm:
            x = (x - self.mean) / self.std
        y, Cc = _nhwc(x), 3
        mods = list(self.features)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                y = _ConvNHWC.apply(y, m.weight, m.bias, 1 if fuse else 0, prec, self._work, flags)
                Cc = m.out_channels
                i += 2 if fuse else 1
            elif isinstance(m, nn.MaxPool2d):
                y = F.max_pool2d(y[..., :Cc].permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1).contiguous()
                i += 1
            elif isinstance(m, nn.BatchNorm2d):
                y = m(y[..., :Cc].permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
                i += 1
            else:                                             # a ReLU that follows a BatchNorm
                y = F.relu(y)
                i += 1
        return _nchw(y, Cc).contiguous()

    def forward(self, x):
        return self._run(self._body, x)
