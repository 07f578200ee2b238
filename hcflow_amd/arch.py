"""Drop-in architecture classes: the reference's module surface over the MI355X engine.

``codes/models/networks.py:36-41`` (define_G) instantiates ``HCFlowNet_SR(opt=opt, step=step)`` /
``HCFlowNet_Rescaling(opt=opt, step=step)`` and the model wrappers call
``netG(hr=, lr=, z=, u=, eps_std=, add_gt_noise=, step=, reverse=, training=)`` by keyword
(HCFlow_SR_model.py:195,208,305,311; HCFlow_Rescaling_model.py:214,219,312,319). The classes here
keep that surface:

* same constructor, same ``forward`` signature and return values
  (HCFlowNet_SR_arch.py:34-75, HCFlowNet_Rescaling_arch.py:26-54);
* an ``nn.Module`` tree whose ``state_dict()`` keys / shapes / order equal the reference's, so
  ``load_network(strict=True)`` (base_model.py:96-120) works on released checkpoints;
* ``named_modules()`` yields ``*ActNorm*`` modules with a writable ``.inited``
  (HCFlow_SR_model.py:462-465).

The modules below only HOLD parameters; all arithmetic runs in hand-written HIP kernels behind the
C ABI of ``include/hcflow.h`` (``libhcflow_hip.so``). There is no PyTorch fallback: without the
library or without a GPU, ``forward`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .config import NetConfig, coupling_io, eps_shapes, opt_get, param_spec


# ------------------------------------------------------------------ parameter containers
def _xavier(shape, scale=0.1):
    w = torch.empty(*shape)
    nn.init.xavier_normal_(w)          # mutil.initialize_weights_xavier (module_util.py:26-43)
    return nn.Parameter(w * scale)


class ActNorm2d(nn.Module):
    """Parameters of ActNorms.ActNorm2d (ActNorms.py:7-27): bias, logs [1,C,1,1] + ``inited``."""

    def __init__(self, num_features, scale=1.):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.logs = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.num_features = num_features
        self.scale = float(scale)
        self.inited = False


class InvertibleConv1x1(nn.Module):
    """Parameter holder of Permutations.InvertibleConv1x1 (Permutations.py:33-58): random orthogonal init; with
    ``LU_decomposed`` the factors of W = P (L o l_mask + I) (U o l_mask^T + diag(sign_s exp(log_s))) -- parameters l, log_s, u,
    buffers p, sign_s (fixed), plain attributes l_mask, eye -- under the reference's names. The engine composes W / W^-1 and
    uses dlogdet = sum(log_s) * pixels (Permutations.py:78-92; hcf_engine_build.inc build_step)."""

    def __init__(self, num_channels, LU_decomposed=False):
        super().__init__()
        w_shape = [num_channels, num_channels]
        w_init = np.linalg.qr(np.random.randn(*w_shape))[0].astype(np.float32)
        if not LU_decomposed:
            self.weight = nn.Parameter(torch.from_numpy(w_init))
        else:
            P, L, U = torch.linalg.lu(torch.from_numpy(w_init).double())     # w = P L U, partial pivoting (scipy.linalg.lu there)
            s = torch.diagonal(U)
            self.register_buffer("p", P.float())
            self.register_buffer("sign_s", torch.sign(s).float())
            self.l = nn.Parameter(L.float())
            self.log_s = nn.Parameter(torch.log(torch.abs(s)).float())
            self.u = nn.Parameter(torch.triu(U, 1).float())
            self.l_mask = torch.tril(torch.ones(*w_shape), -1)
            self.eye = torch.eye(*w_shape)
        self.w_shape = w_shape
        self.LU = LU_decomposed


class Conv2d(nn.Module):
    """Basic.Conv2d with do_actnorm=True (Basic.py:14-53): bias-free conv weight + ActNorm2d."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = _xavier((cout, cin, k, k))
        self.actnorm = ActNorm2d(cout)


class Conv2dZeros(nn.Module):
    """Basic.Conv2dZeros (Basic.py:57-72): zero-initialised weight, bias, logs [cout,1,1]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.logs = nn.Parameter(torch.zeros(cout, 1, 1))


class PlainConv(nn.Module):
    """nn.Conv2d(cin, cout, 3, 1, 1, bias=True) parameters."""

    def __init__(self, cin, cout, init="xavier"):
        super().__init__()
        if init == "xavier":
            self.weight = _xavier((cout, cin, 3, 3))
            self.bias = nn.Parameter(torch.zeros(cout))
        elif init == "zero":
            self.weight = nn.Parameter(torch.zeros(cout, cin, 3, 3))
            self.bias = nn.Parameter(torch.zeros(cout))
        else:                                   # torch's default Conv2d reset_parameters
            ref = nn.Conv2d(cin, cout, 3, 1, 1, bias=True)
            self.weight = nn.Parameter(ref.weight.detach().clone())
            self.bias = nn.Parameter(ref.bias.detach().clone())


class FCN(nn.Module):
    """Basic.FCN (Basic.py:426-447)."""

    def __init__(self, cin, cout, hidden):
        super().__init__()
        self.conv1 = Conv2d(cin, hidden, 3)
        self.conv2 = Conv2d(hidden, hidden, 1)
        self.conv3 = Conv2dZeros(hidden, cout)


class DenseBlock(nn.Module):
    """Basic.DenseBlock, for_flow=True (Basic.py:329-347): conv5 zero-initialised."""

    def __init__(self, cin, cout, gc):
        super().__init__()
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), PlainConv(cin + i * gc, gc))
        self.conv5 = PlainConv(cin + 4 * gc, cout, init="zero")


class ResidualDenseBlock(nn.Module):
    """Basic.ResidualDenseBlock (Basic.py:360-377)."""

    def __init__(self, nf, gc):
        super().__init__()
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), PlainConv(nf + i * gc, gc))
        self.conv5 = PlainConv(nf + 4 * gc, nf)


class RRDB(nn.Module):
    """Basic.RRDB (Basic.py:387-392)."""

    def __init__(self, nf, gc):
        super().__init__()
        self.RDB1 = ResidualDenseBlock(nf, gc)
        self.RDB2 = ResidualDenseBlock(nf, gc)
        self.RDB3 = ResidualDenseBlock(nf, gc)


class AffineCoupling(nn.Module):
    """AffineCouplings.AffineCoupling / AffineCoupling3shift parameter holder (``f``)."""

    def __init__(self, C, cond, coupling, nn_module, hidden, lr_vs_others=True):
        super().__init__()
        fin, fout = coupling_io(C, cond, coupling, lr_vs_others)
        self.f = FCN(fin, fout, hidden) if nn_module == "FCN" else DenseBlock(fin, fout, hidden)


class FlowStep(nn.Module):
    """FlowStep (FlowStep.py:8-38): actnorm, permute, affine."""

    def __init__(self, C, cond, perm, coupling, nn_module, hidden, lr_vs_others=True, LU_decomposed=False):
        super().__init__()
        self.actnorm = ActNorm2d(C)
        if perm == "invconv":
            self.permute = InvertibleConv1x1(C, LU_decomposed=LU_decomposed)
        else:
            self.permute = None
        self.affine = AffineCoupling(C, cond, coupling, nn_module, hidden, lr_vs_others)


class SqueezeLayer(nn.Module):
    def __init__(self, factor=2):
        super().__init__()
        self.factor = factor


class HaarDownsampling(nn.Module):
    """Basic.HaarDownsampling (Basic.py:450-468): frozen +-1 ``haar_weights`` parameter."""

    def __init__(self, channel_in):
        super().__init__()
        w = torch.ones(4, 1, 2, 2)
        w[1, 0, 0, 1] = -1
        w[1, 0, 1, 1] = -1
        w[2, 0, 1, 0] = -1
        w[2, 0, 1, 1] = -1
        w[3, 0, 1, 0] = -1
        w[3, 0, 0, 1] = -1
        self.haar_weights = nn.Parameter(torch.cat([w] * channel_in, 0), requires_grad=False)


class Split(nn.Module):
    def __init__(self, num_channels_split, level):
        super().__init__()
        self.num_channels_split = num_channels_split
        self.level = level


class ConditionalFlow(nn.Module):
    """ConditionalFlow (ConditionalFlow.py:15-41)."""

    def __init__(self, cfg: NetConfig, level: int):
        super().__init__()
        C, ns = cfg.level_channels(level), cfg.split_channels(level)
        cin = ns + cfg.cond_ch * cfg.num_levels_condition(level)
        self.conv_first = PlainConv(cin, cfg.rrdb_nf, init="default")
        self.RRDB_trunk0 = nn.Sequential(*[RRDB(cfg.rrdb_nf, cfg.rrdb_gc) for _ in range(cfg.rrdb_nb[0])])
        self.RRDB_trunk1 = nn.Sequential(*[RRDB(cfg.rrdb_nf, cfg.rrdb_gc) for _ in range(cfg.rrdb_nb[1])])
        self.trunk_conv1 = PlainConv(cfg.rrdb_nf, cfg.rrdb_nf, init="default")
        self.additional_flow_steps = nn.ModuleList(
            [FlowStep(C - ns, cfg.cond_ch, cfg.c_perm, cfg.c_coupling, cfg.c_nn_module, cfg.c_hidden, LU_decomposed=cfg.lu)
             for _ in range(cfg.after[level])])
        self.f = Conv2dZeros(cfg.cond_ch, (C - ns) * 2)


class FlowNet(nn.Module):
    """FlowNet.__init__ of FlowNet_SR_x4 / FlowNet_SR_x8 / FlowNet_Rescaling_x4 (layer list only)."""

    def __init__(self, cfg: NetConfig, hr_size: int = 160):
        super().__init__()
        self.layers = nn.ModuleList()
        self.output_shapes = []
        H = W = hr_size
        C = cfg.in_nc
        for level in range(cfg.L):
            self.layers.append(HaarDownsampling(C) if cfg.squeeze == "haar" else SqueezeLayer(2))
            C, H, W = C * 4, H // 2, W // 2
            self.output_shapes.append([-1, C, H, W])
            for k in range(cfg.K[level] - cfg.after[level]):
                lrv = True if cfg.sr else (k % 2 == 0)
                self.layers.append(FlowStep(C, 0, cfg.perm, cfg.coupling, cfg.nn_module, cfg.hidden, lrv, LU_decomposed=cfg.lu))
                self.output_shapes.append([-1, C, H, W])
            ns = cfg.split_channels(level)
            self.layers.append(Split(ns, level))
            setattr(self, "level%d_condFlow" % level, ConditionalFlow(cfg, level))
            C = ns
            self.output_shapes.append([-1, C, H, W])
        self.H, self.W = H, W
        print('shapes:', self.output_shapes)      # FlowNet_SR_x4.py:71 (part of the boundary's observable behaviour)


# ------------------------------------------------------------------ NLL training step (autograd bridge)
class _SRNLLStep(torch.autograd.Function):
    """``_, nll = netG(hr=, lr=, reverse=False)`` with gradients (HCFlow_SR_model.py:195-199): the engine keeps the
    intermediate tensors of the forward pass in HBM (hcf_train_forward_sr) and produces d nll / d parameters for the
    whole net in one call (hcf_train_backward); this Function only hands the flat gradient buffer to autograd."""

    @staticmethod
    def forward(ctx, module, hr, lr, noise, *params):
        dev = hr.device
        eng, idx = module._engine_for(dev)
        B, _, H, W = hr.shape
        s = module.cfg.scale
        out_lr = torch.empty(B, 3, H // s, W // s, device=dev)
        nll = torch.empty(1, device=dev)
        logdet = torch.empty(B, device=dev)
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_forward_sr(eng.handle, hr.data_ptr(), lr.data_ptr(), noise.data_ptr(),
                                              out_lr.data_ptr(), nll.data_ptr(), logdet.data_ptr(), B, H, W,
                                              module._stream(idx))
        _lib.check(rc, eng.handle, "hcf_train_forward_sr")
        ctx.eng, ctx.idx = eng, idx
        ctx.keep = (hr, lr, noise)                       # the engine's tape holds raw pointers to these
        ctx.meta = [(tuple(p.shape), p.numel(), bool(p.requires_grad)) for p in params]
        ctx.mark_non_differentiable(out_lr, logdet)
        return out_lr, nll.view(()), logdet

    @staticmethod
    def backward(ctx, g_lr, g_nll, g_logdet):
        eng, idx = ctx.eng, ctx.idx
        total = sum(n for _, n, _ in ctx.meta)
        flat = torch.empty(total, device=ctx.keep[0].device, dtype=torch.float32)
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_backward(eng.handle, float(g_nll), flat.data_ptr(), total,
                                            C.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
        _lib.check(rc, eng.handle, "hcf_train_backward")
        grads, off = [], 0
        for shape, n, need in ctx.meta:
            grads.append(flat[off:off + n].view(shape) if need else None)
            off += n
        return (None, None, None, None) + tuple(grads)


# ---- the same step as TWO autograd nodes (gradient all-reduce overlapped with the backward pass) -----------------------------
# The reference trains under DistributedDataParallel (HCFlow_SR_model.py:33-36): DDP reduces a bucket of gradients as soon as its
# parameters' AccumulateGrad hooks have fired, under the rest of the backward pass. One autograd node for the whole net hands over
# every gradient at once, after the last kernel. Two nodes: the OUTER one owns the parameters whose gradients are final after the first
# phase of the engine's backward pass (hcf_train_backward_phase: the level-0 conditional flow, about half of an SR x4 net), the
# INNER one the rest; autograd runs outer.backward (phase 0: its gradients go to DDP, whose all-reduce starts on its own stream), then
# inner.backward (phase 1). `state` is any object with forward() -> tuple of outputs, backward(phase, grad_outputs) -> list of gradients
# (phase 0: of `early`, phase 1: of `late`): the engine-backed one below, a plain-torch one in tests/test_dist_cpu.py.
class _LateNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *late):
        ctx.state = state
        return torch.zeros((), device=late[0].device if late else None)

    @staticmethod
    def backward(ctx, g_token):
        return (None,) + tuple(ctx.state.backward(1, None))


class _EarlyNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, token, *early):
        ctx.state = state
        outs = state.forward()
        ctx.mark_non_differentiable(*[o for o, d in zip(outs, state.differentiable) if not d])
        return outs

    @staticmethod
    def backward(ctx, *g_outs):
        grads = ctx.state.backward(0, g_outs)
        return (None, torch.zeros(()).to(g_outs[0].device if g_outs[0] is not None else "cpu")) + tuple(grads)


def two_phase_apply(state, early, late):
    """outputs of ``state.forward()`` on an autograd graph of two nodes (see above); ``early`` / ``late``: the parameter tensors whose
    gradients ``state.backward(0, ...)`` / ``state.backward(1, ...)`` return."""
    token = _LateNode.apply(state, *late)
    return _EarlyNode.apply(state, token, *early)


class _SRNLLTwoPhase:
    """Engine-backed state of two_phase_apply for the NLL step: hcf_train_forward_sr, then hcf_train_backward_phase 0 / 1 into ONE
    flat gradient buffer handed out as views."""
    differentiable = (False, True, False)

    def __init__(self, module, hr, lr, noise, params, early_idx):
        self.module, self.hr, self.lr, self.noise = module, hr, lr, noise
        self.meta = [(tuple(p.shape), p.numel(), bool(p.requires_grad)) for p in params]
        self.early_idx = set(early_idx)
        self.flat = None

    def forward(self):
        m, hr = self.module, self.hr
        dev = hr.device
        self.eng, self.idx = m._engine_for(dev)
        B, _, H, W = hr.shape
        s = m.cfg.scale
        out_lr = torch.empty(B, 3, H // s, W // s, device=dev)
        nll = torch.empty(1, device=dev)
        logdet = torch.empty(B, device=dev)
        eng = self.eng
        with torch.cuda.device(self.idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_forward_sr(eng.handle, hr.data_ptr(), self.lr.data_ptr(), self.noise.data_ptr(),
                                              out_lr.data_ptr(), nll.data_ptr(), logdet.data_ptr(), B, H, W, m._stream(self.idx))
        _lib.check(rc, eng.handle, "hcf_train_forward_sr")
        return out_lr, nll.view(()), logdet

    def backward(self, phase, g_outs):
        eng, idx = self.eng, self.idx
        total = sum(n for _, n, _ in self.meta)
        if phase == 0:
            self.flat = torch.empty(total, device=self.hr.device, dtype=torch.float32)
            self.g_nll = float(g_outs[1])
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_backward_phase(eng.handle, phase, self.g_nll, self.flat.data_ptr(), total,
                                                  C.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
        _lib.check(rc, eng.handle, "hcf_train_backward_phase")
        grads, off = [], 0
        for i, (shape, n, need) in enumerate(self.meta):
            if (i in self.early_idx) == (phase == 0):
                grads.append(self.flat[off:off + n].view(shape) if need else None)
            off += n
        return grads


def _grad_nodes() -> int:
    """Autograd nodes of the NLL step: 1 (default) or 2 (HCFLOW_GRAD_NODES=2). Measured on one MI355X (config 5, B = 16,
    profiles/r06_notes.md): the two-phase backward costs 6-9 ms of a 57.7 ms step -- its mid-pass flush joins the low-priority
    weight-gradient stream, which runs ~10 ms behind the data-gradient chain and otherwise catches up in the chain's gaps -- while
    the all-reduce it hides is 92.9 MB, ~2-3 ms on xGMI. It pays only where the gradient all-reduce is slower than that (PCIe or
    network-attached ranks); over xGMI the one-node step with the all-reduce behind the last kernel is the faster one."""
    return 2 if os.environ.get("HCFLOW_GRAD_NODES") == "2" else 1


class _SRReverseStep(torch.autograd.Function):
    """``fake_H = netG(lr=, eps_std=, reverse=True)`` with gradients w.r.t. the parameters (the HR pixel / feature /
    GAN losses of the HCFlow+ / ++ recipes, HCFlow_SR_model.py:207-255): hcf_train_inverse keeps the tape,
    hcf_train_backward_inverse turns dL/d fake_H into the flat parameter gradient."""

    @staticmethod
    def forward(ctx, module, lr, tau, seed, clamp, eps, *params):
        dev = lr.device
        eng, idx = module._engine_for(dev)
        cfg = module.cfg
        B, _, h, w = lr.shape
        out = torch.empty(B, 3, h * cfg.scale, w * cfg.scale, device=dev, dtype=torch.float32)
        shapes = eps_shapes(cfg, B, h, w)
        arr = (C.c_void_p * len(shapes))()
        keep = []
        if eps is not None:
            assert len(eps) == len(shapes)
            for i, (e, s) in enumerate(zip(eps, shapes)):
                if e is None:
                    arr[i] = None
                    continue
                e = module._prep(e, dev)
                assert tuple(e.shape) == tuple(s), (tuple(e.shape), s)
                keep.append(e)
                arr[i] = e.data_ptr()
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 1), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_inverse(eng.handle, lr.data_ptr(), arr, len(shapes), float(tau), int(seed),
                                           out.data_ptr(), B, h, w, 0 if clamp else _lib.FLAG_NO_CLAMP,
                                           module._stream(idx))
        _lib.check(rc, eng.handle, "hcf_train_inverse")
        ctx.eng, ctx.idx = eng, idx
        ctx.keep = (lr, keep)
        ctx.lr_needs_grad = bool(lr.requires_grad)
        ctx.meta = [(tuple(p.shape), p.numel(), bool(p.requires_grad)) for p in params]
        return out

    @staticmethod
    def backward(ctx, g_out):
        eng, idx = ctx.eng, ctx.idx
        total = sum(n for _, n, _ in ctx.meta)
        g_out = g_out.to(torch.float32).contiguous()
        flat = torch.empty(total, device=g_out.device, dtype=torch.float32)
        g_lr = torch.empty_like(ctx.keep[0]) if ctx.lr_needs_grad else None
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 1), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_backward_inverse(eng.handle, g_out.data_ptr(), flat.data_ptr(), total,
                                                    None if g_lr is None else g_lr.data_ptr(),
                                                    C.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
        _lib.check(rc, eng.handle, "hcf_train_backward_inverse")
        grads, off = [], 0
        for shape, n, need in ctx.meta:
            grads.append(flat[off:off + n].view(shape) if need else None)
            off += n
        return (None, g_lr, None, None, None, None) + tuple(grads)


class _RescaleForwardStep(torch.autograd.Function):
    """``fake_LR, z1, z2 = netG(hr=, reverse=False)`` of the rescaling net with gradients
    (HCFlow_Rescaling_model.optimize_parameters, :212-216): tape slot 0 (the inverse pass of the same step uses slot 1)."""

    @staticmethod
    def forward(ctx, module, hr, clamp, *params):
        dev = hr.device
        eng, idx = module._engine_for(dev)
        cfg = module.cfg
        B, _, H, W = hr.shape
        out_lr = torch.empty(B, 3, H // 4, W // 4, device=dev)
        z1 = torch.empty(B, cfg.level_channels(0) - cfg.split_channels(0), H // 2, W // 2, device=dev)
        z2 = torch.empty(B, cfg.level_channels(1) - cfg.split_channels(1), H // 4, W // 4, device=dev)
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_forward_rescale(eng.handle, hr.data_ptr(), out_lr.data_ptr(), z1.data_ptr(), z2.data_ptr(),
                                                   B, H, W, 0 if clamp else _lib.FLAG_NO_CLAMP, module._stream(idx))
        _lib.check(rc, eng.handle, "hcf_train_forward_rescale")
        ctx.eng, ctx.idx, ctx.keep = eng, idx, hr
        ctx.meta = [(tuple(p.shape), p.numel(), bool(p.requires_grad)) for p in params]
        return out_lr, z1, z2

    @staticmethod
    def backward(ctx, g_lr, g_z1, g_z2):
        eng, idx = ctx.eng, ctx.idx
        total = sum(n for _, n, _ in ctx.meta)
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_lr, g_z1, g_z2)]
        flat = torch.empty(total, device=ctx.keep.device, dtype=torch.float32)
        with torch.cuda.device(idx):
            _lib.check(eng.lib.hcf_train_select_tape(eng.handle, 0), eng.handle, "hcf_train_select_tape")
            rc = eng.lib.hcf_train_backward_rescale(eng.handle, *[None if g is None else g.data_ptr() for g in gs],
                                                    flat.data_ptr(), total,
                                                    C.c_void_p(torch.cuda.current_stream(idx).cuda_stream))
        _lib.check(rc, eng.handle, "hcf_train_backward_rescale")
        grads, off = [], 0
        for shape, n, need in ctx.meta:
            grads.append(flat[off:off + n].view(shape) if need else None)
            off += n
        return (None, None, None) + tuple(grads)


_POOL = []
_POOL_LOCK = threading.Lock()


def _enqueue_pool():
    """One helper thread per process that enqueues the second half batch of a split inference call (see _run_checked)."""
    if not _POOL:
        with _POOL_LOCK:                 # (nn.DataParallel runs its replicas on parallel threads: one executor, not one per first caller)
            if not _POOL:
                from concurrent.futures import ThreadPoolExecutor
                _POOL.append(ThreadPoolExecutor(max_workers=1, thread_name_prefix="hcflow-enqueue"))
    return _POOL[0]


def _capturing() -> bool:
    """torch.cuda.is_current_stream_capturing() that answers False on a host without a GPU (the call then fails loudly further down)."""
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class _StaleParameters(Exception):
    """A call that was enqueued before its parameter stamp was verified (``_engine_for(defer=True)``) found the parameters changed:
    the enqueued pass has been drained and the call is redone on the checked path."""


def _split_threaded() -> bool:
    """Second half of a split call enqueued by the helper thread? Default: yes (Face x8 at LR 20 x 20: 1 542 against 1 403 img/s with
    one thread enqueueing both halves, 1 416 unsplit; the -25 % once measured for this case was a garbage-collection pause inside
    an 8-call timing loop). HCF_SPLIT_THREADED=0 turns it off (A/B knob)."""
    return os.environ.get("HCF_SPLIT_THREADED", "1") == "1"


# ------------------------------------------------------------------ engine-backed top modules
class _EngineModule(nn.Module):
    """Shared plumbing: parameter upload / repack tracking and raw-pointer calls into the C ABI."""

    def _setup(self, opt):
        self.opt = opt
        self.cfg = NetConfig.from_opt(opt)
        hr_size = opt_get(opt, ['datasets', 'train', 'GT_size'], 160)
        self.flow = FlowNet(self.cfg, hr_size)
        spec = [(k, tuple(s)) for k, s, _ in param_spec(self.cfg)]
        have = [(k, tuple(v.shape)) for k, v in self.state_dict().items()]
        assert have == spec, "internal: module tree does not match the reference state_dict table"
        object.__setattr__(self, "_spec_keys", [k for k, _ in spec])
        # engines are per device (nn.DataParallel replicas share this dict but not the entries)
        object.__setattr__(self, "_engines", {})
        # conv numerics. Default "f16x3": fp32-equivalent split products on the f16 matrix cores with fp32 accumulation
        # (deviation from an fp64 evaluation of the full nets equals plain fp32's; inputs beyond the f16 range
        # are detected and the pass is re-run exactly). HCFLOW_PRECISION=exact (or set_precision("exact")) selects the fp32
        # MFMA kernels, a bit-exact fp32 fma chain at 1/3 of the throughput.
        default_prec = os.environ.get("HCFLOW_PRECISION", "f16x3")
        assert default_prec in _lib.Engine.PRECISIONS, "HCFLOW_PRECISION must be one of %s" % list(_lib.Engine.PRECISIONS)
        object.__setattr__(self, "_precision", [default_prec])
        # f16x3 range check: "sync" (default) asks the engine after every pass and re-runs an overflowed pass exactly
        # (one host-device synchronisation per call, here in Python, never inside the C ABI); "lazy" leaves it to
        # check_range() (the calls only enqueue: CUDA-graph capturable); "off" skips the read-back altogether.
        object.__setattr__(self, "_range_check", [os.environ.get("HCFLOW_RANGE_CHECK", "sync")])
        object.__setattr__(self, "_cond_key", {})
        # DEFAULT since round 5 (HCFLOW_STREAMS=1 or set_streams(1) turns it off): inference calls of >= 4 samples run as TWO half
        # batches on the process' two side streams (hcf_aux_stream; two engines: own workspace and packs, the same parameter
        # tensors), the second half enqueued by a helper thread while this one enqueues the first: every op of the path is
        # per-sample, the convolutions are persistent one-block-per-CU launches, and the second stream's kernels fill the ragged
        # last rounds and launch boundaries of the first's. Same box, GC-quiet loops (profiles/r05_notes.md sections 4, 10):
        # config 2 +3-4 %, config 4 +8-9 %, Face x8 +9 %; B = 1 calls and the training pass are unaffected (the side streams are
        # the two the training pass uses as well). Overlapping kernels void per-kernel durations (HIP events, rocprofv3): bench.py
        # takes its roofline block from a single-stream leg.
        object.__setattr__(self, "_nstreams", [max(1, min(2, int(os.environ.get("HCFLOW_STREAMS", "2"))))])
        object.__setattr__(self, "_side_streams", {})

    def set_streams(self, n: int):
        """2 (default): inference calls of >= 4 samples are split into two half batches that run side by side on the process' two
        side streams (joined before the call returns); 1: every call runs on the caller's stream with one engine."""
        assert n in (1, 2), n
        self._nstreams[0] = int(n)
        return self

    def _side_stream(self, idx):
        """Side stream ``idx = (device, slot)``: the PROCESS' pool of two per device (include/hcflow.h: hcf_aux_stream), shared with the
        engines' training passes, so that a process never holds more than the caller's stream + two (HIP spreads streams over four
        hardware queues; with streams of its own for the split calls a process that also trained ran its backward pass at 66
        instead of 37 ms)."""
        st = self._side_streams.get(idx)
        if st is None:
            dev, slot = idx
            h = C.c_void_p()
            rc = _lib.load().hcf_aux_stream(int(dev), int(slot), C.byref(h))
            if rc != 0 or not h.value:
                raise _lib.HcfError("hcf_aux_stream(%d, %d) failed (%d)" % (dev, slot, rc))
            st = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", dev))
            self._side_streams[idx] = st
        return st

    def engines(self):
        """Every live engine of this module's device (the primary and, once a split call has run, its twin)."""
        dev = self._device()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        self._engine_for(dev)
        return [ent["engine"] for k, ent in self._engines.items() if k == idx or (isinstance(k, tuple) and k[0] == idx)]

    def set_range_check(self, mode: str):
        assert mode in ("sync", "lazy", "off"), mode
        self._range_check[0] = mode
        return self

    def check_range(self) -> bool:
        """"lazy" mode: True if an f16x3 pass since the last check overflowed the f16 range (its outputs are invalid and
        must be recomputed with set_precision("exact")). Waits for the enqueued passes."""
        return any([ent["engine"].check_range() for ent in self._engines.values()])   # every engine's flag is read AND cleared

    def invalidate(self):
        """Force a repack of every engine on its next call. Needed after writes the version counters do not see
        (``p.data.copy_()`` / ``p.data.mul_()`` as EMA or clipping code does): the engine otherwise keeps its packed
        weights, inverse matrices and log-det constants. Also re-arms the "ActNorm already fitted" bookkeeping."""
        self.__dict__.pop("_slots", None)             # re-resolve the parameter slots (a sub-module may have been replaced)
        for ent in self._engines.values():
            ent["stamp"] = None
            ent["ptrs"] = None
        for m in self.modules():
            if isinstance(m, ActNorm2d) and hasattr(m, "_hcf_fitted"):
                del m._hcf_fitted
        self._cond_key.clear()
        return self

    def _tensors(self):
        """[(key, tensor)] in state_dict order, resolved through the ATTRIBUTE tree: nn.DataParallel replicas carry their
        parameters as plain tensor attributes (torch >= 1.5: replica.parameters() is empty), and those tensors are the
        ones autograd must see so that gradients flow back to the wrapped module (HCFlow_SR_model.py:33-36).
        The (owner module, attribute) pairs are resolved once per module OBJECT (walking 1 500-1 900 dotted paths costs
        ~10 ms of Python per call, several times per forward); a replica is a different object and resolves its own."""
        return list(zip(self._spec_keys, self._tensor_list()))

    def _tensor_list(self):
        """The tensors of ``_tensors()`` without their keys (the per-call stamp check walks only this: 1 500-1 900 dictionary
        look-ups; under the default `sync` range policy the Python in front of a call's first launch is GPU idle time)."""
        out = []
        for obj, name in self._slots_for_self():
            t = obj._parameters.get(name)
            out.append(t if t is not None else getattr(obj, name))
        return out

    @staticmethod
    def _stamp_of(tens):
        """(addresses, versions) of the parameter tensors as TWO flat lists of ints: 1 500-1 900 (ptr, version) tuples per call were
        1 500-1 900 GC-tracked containers per call, i.e. a generation-0 collection every other call and, every few dozen calls, a
        full collection over the process' ~200 k objects -- a 50-70 ms pause in front of a 13 ms call (profiles/r05_notes.md 10)."""
        return ([p.data_ptr() for p in tens], [p._version for p in tens])

    def _slots_for_self(self):
        cached = self.__dict__.get("_slots")
        if cached is not None and cached[0] == id(self):
            # a replaced sub-module (EMA swap, a flow block exchanged by assignment) must not leave the engine bound to the old
            # module's tensors: every (parent, attribute, child) edge of the resolved paths is re-checked (~600 dict lookups)
            for parent, name, child in cached[2]:
                if parent._modules.get(name) is not child:
                    break
            else:
                return cached[1]
        slots, edges, seen = [], [], set()
        for key in self._spec_keys:
            obj = self
            path = key.split(".")
            for a in path[:-1]:
                nxt = getattr(obj, a)
                if (id(obj), a) not in seen:
                    seen.add((id(obj), a))
                    edges.append((obj, a, nxt))
                obj = nxt
            slots.append((obj, path[-1]))
        object.__setattr__(self, "_slots", (id(self), slots, edges))
        return slots

    def _device(self):
        obj, name = self._slots_for_self()[0]
        t = obj._parameters.get(name)
        return (t if t is not None else getattr(obj, name)).device

    def set_precision(self, mode: str):
        """"exact": fp32 MFMA convolutions (default). "f16x3": fp32-equivalent split products on the
        f16 matrix cores with fp32 accumulation (include/hcflow.h: hcf_set_precision)."""
        assert mode in _lib.Engine.PRECISIONS, mode
        self._precision[0] = mode
        for ent in self._engines.values():
            ent["engine"].set_precision(mode)
        return self

    # -- engine management
    def _engine_for(self, device: torch.device, slot: int = 0, same_call: bool = False, defer: bool = False,
                    twin_hint: bool = False):
        """The engine of ``device`` (slot 0), or its twin (slot 1: the second half batch of a split inference call runs on it,
        beside slot 0's, on a second HIP stream -- its own workspace and packs, the same parameter tensors).

        ``defer`` (eval-mode inference calls): the walk over the 1 500-1 900 parameter tensors that tells whether the packs are
        current (~1 ms of Python) is GPU idle time when it runs in front of a call's first launch -- under the default `sync`
        policy every call starts on an idle GPU. An engine whose last check found the parameters unchanged is handed out
        unchecked; ``_run_checked`` verifies the stamp AFTER the pass is enqueued, while the GPU runs it, and on a mismatch drains the
        pass (it only wrote the workspace and the output) and has the call redone on the checked path (``_StaleParameters``). The
        pass never reads the caller's parameter tensors, only the engine's packs."""
        if device.type != "cuda":
            raise _lib.HcfError(
                "hcflow_amd runs on MI355X only: move the module and its inputs to a GPU "
                "(module parameters are on %s). There is no CPU fallback." % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = idx if slot == 0 else (idx, slot)
        ent = self._engines.get(key)
        if ent is None:
            ent = {"engine": _lib.Engine(self.cfg), "stamp": None}
            ent["engine"].set_precision(self._precision[0])
            self._engines[key] = ent
        if slot == 0:
            self.__dict__["_call_stamp"] = None
            self.__dict__["_deferred"] = None
            self.__dict__["_ran_unverified"] = False
        if (defer and slot == 0) or (same_call and self.__dict__.get("_deferred") is not None):
            if ent["stamp"] is not None and ent.get("stable"):
                if slot == 0:
                    self.__dict__["_deferred"] = []
                    self.__dict__["_ran_unverified"] = True      # until _run_checked has compared the stamps
                self.__dict__["_deferred"].append(ent)
                return ent["engine"], idx
        # (same_call: the twin engine of a split call is looked up right after the primary one -- the ~1 ms walk over the 1 500-1 900
        #  parameter tensors is not repeated)
        memo = self.__dict__.get("_call_stamp") if same_call else None
        if memo is not None:
            tens, stamp = memo
        else:
            tens = self._tensor_list()
            stamp = self._stamp_of(tens)
            self.__dict__["_call_stamp"] = (tens, stamp)
        ent["stable"] = ent["stamp"] == stamp        # unchanged since the last check: the next eval-mode call may defer its own
        if ent["stamp"] != stamp:
            named = list(zip(self._spec_keys, tens))
            eng = ent["engine"]
            self._cond_key.pop(idx, None)
            ptrs = tuple(p.data_ptr() for _, p in named)
            on_dev = all(p.device.type == "cuda" and p.device.index == idx and p.dtype == torch.float32 and p.is_contiguous()
                         for _, p in named)
            # The device-side refresh re-reads every PARAMETER; the LU-decomposed invertible convs' fixed buffers (p, sign_s:
            # Permutations.py:54-55) are captured by hcf_finalize only. A load_state_dict / in-place write that changes them
            # (a checkpoint's pivoting differs from the random init's) takes the full host path below.
            # Compared by CONTENT when the cheap (address, version) stamp moved: nn.DataParallel hands replicas on devices >= 1
            # freshly broadcast buffer tensors at every forward -- by address alone each of their calls took the ~1.7 s host path.
            lu_t = [p for key, p in named if key.endswith(".permute.p") or key.endswith(".permute.sign_s")]
            lu_stamp = tuple((p.data_ptr(), p._version) for p in lu_t)
            lu_same = ent.get("lu_stamp") == lu_stamp
            if not lu_same and ent.get("lu_copy") is not None and len(ent["lu_copy"]) == len(lu_t):
                lu_same = all(a.shape == b.shape and a.device == b.device and bool(torch.equal(a, b)) for a, b in zip(ent["lu_copy"], lu_t))
            ent["lu_stamp"] = lu_stamp
            if not lu_same or ent.get("lu_copy") is None:
                ent["lu_copy"] = [p.detach().clone() for p in lu_t]
            if on_dev and lu_same and ent.get("ptrs") == ptrs:
                # same tensors, new contents (optimiser step): rewrite the packs on the device
                with torch.cuda.device(idx):
                    eng.refresh_from_device(self._stream(idx))
            elif on_dev and lu_same and ent.get("ptrs") is not None:
                # other tensors of the same shapes (nn.DataParallel replicas are re-created every forward,
                # HCFlow_SR_model.py:33-36 in the non-distributed case): re-bind and refresh on the device
                with torch.cuda.device(idx):
                    for key, p in named:
                        eng.bind_param_device(key, p.data_ptr())
                    eng.refresh_from_device(self._stream(idx))
                ent["ptrs"] = ptrs
            else:
                cpu = [(key, t.detach().to("cpu", torch.float32).contiguous()) for key, t in named]

                def host_build(e_):
                    for key, t in cpu:
                        e_.set_param(key, t)
                    e_.finalize(idx)                              # (host-side packing: 1.7 s for the SR x4 net)
                    if on_dev:
                        for key, p in named:
                            e_.bind_param_device(key, p.data_ptr())
                # twin_hint (slot 0 of a call that will be split): a twin that has never been built is built BESIDE this engine, on
                # the helper thread -- the first batched call costs one engine build (1.9 s), not two in a row (3.7 s)
                twin, twin_fut = None, None
                if twin_hint and slot == 0:
                    twin = self._engines.get((idx, 1))
                    if twin is None:
                        twin = {"engine": _lib.Engine(self.cfg), "stamp": None}
                        twin["engine"].set_precision(self._precision[0])
                        self._engines[(idx, 1)] = twin
                    if twin["stamp"] is None and twin.get("ptrs") is None:
                        twin_fut = _enqueue_pool().submit(host_build, twin["engine"])
                    else:
                        twin = None
                host_build(eng)
                ent["ptrs"] = ptrs if on_dev else None
                if twin_fut is not None:
                    twin_fut.result()
                    twin["ptrs"] = ptrs if on_dev else None
                    twin["lu_stamp"] = lu_stamp
                    twin["stamp"] = stamp
                    twin["stable"] = False
            ent["stamp"] = stamp
        return ent["engine"], idx

    def _wants_grad(self):
        return torch.is_grad_enabled() and any(p.requires_grad for _, p in self._tensors())

    def _params(self):
        return [p for _, p in self._tensors()]

    def _run_checked(self, eng, idx, call, what, batch=0, call_sample=None, parts=None, threaded=False):
        """One inference pass through the C ABI under the range-check policy. ``call(eng, lo, hi, stream)`` enqueues samples
        [lo, hi) on ``eng`` and returns the status. ``parts`` (optional): [(engine, lo, hi, torch stream or None)] -- the split of a
        batch over the device's engines / streams (None = the caller's stream); the side streams are joined before returning.
        ``call_sample(b)`` (optional) enqueues sample ``b`` alone on ``eng``: every op of the path is per-sample
        (HCFlowNet_SR_arch.py:70-75), so an activation beyond the f16 range costs an exact re-run of the samples whose tiles saw
        it (B = 1 passes on the fp32-MFMA kernels into the same output rows), not of the whole batch."""
        with torch.cuda.device(idx):
            cur = torch.cuda.current_stream(idx)
            if parts is None:
                parts = [(eng, 0, batch, None)]
            if len(parts) == 1:
                e_, lo, hi, _ = parts[0]
                _lib.check(call(e_, lo, hi, C.c_void_p(cur.cuda_stream)), e_.handle, what)
            else:
                # two half batches on two side streams. ``threaded`` (large samples): the second half's launches are enqueued by a
                # helper thread WHILE this thread enqueues the first half's (ctypes drops the GIL inside the C call): config 2
                # +4.5 % instead of +2.8 %, Face x8 +9 % (profiles/r05_notes.md sections 4, 11); HCF_SPLIT_THREADED=0: enqueued in turn
                for _, _, _, st_ in parts:
                    st_.wait_stream(cur)                          # the inputs were produced on the caller's stream
                (e0, lo0, hi0, s0), (e1, lo1, hi1, s1) = parts
                if threaded and torch.cuda.is_current_stream_capturing():
                    threaded = False                              # a capturing stream's launches stay on the capturing thread
                if threaded:
                    fut = _enqueue_pool().submit(call, e1, lo1, hi1, C.c_void_p(s1.cuda_stream))
                    rc0 = call(e0, lo0, hi0, C.c_void_p(s0.cuda_stream))
                    rc1 = fut.result()
                else:
                    rc0 = call(e0, lo0, hi0, C.c_void_p(s0.cuda_stream))
                    rc1 = call(e1, lo1, hi1, C.c_void_p(s1.cuda_stream))
                _lib.check(rc0, e0.handle, what)
                _lib.check(rc1, e1.handle, what)
                for _, _, _, st_ in parts:
                    cur.wait_stream(st_)                          # joined: the output is complete on the caller's stream
            deferred = self.__dict__.get("_deferred")
            if deferred:
                # the stamp check this call skipped in front of its first launch (_engine_for(defer=True)), now beside the running pass
                self.__dict__["_deferred"] = None
                memo = self.__dict__.get("_call_stamp")
                stamp = memo[1] if memo is not None else self._stamp_of(self._tensor_list())
                if any(ent["stamp"] != stamp for ent in deferred):
                    for ent in deferred:
                        ent["stable"] = False
                    cur.synchronize()                             # the stale pass only wrote the workspace and `out`
                    for e_, _, _, _ in parts:
                        e_.check_range_samples()                  # (whatever it flagged is dropped with it)
                    raise _StaleParameters()
                self.__dict__["_ran_unverified"] = False          # verified: from here on an error is the call's own
            if self._precision[0] != "f16x3" or self._range_check[0] != "sync":
                return
            flagged, any_over = [], False
            for e_, lo, hi, _ in parts:
                over, slots = e_.check_range_samples()
                any_over = any_over or over
                if over:
                    flagged += [lo + b for b in range(hi - lo) if (slots >> (b % 30)) & 1]
            if not any_over:
                return
            eng.set_precision("exact")                       # an activation left the f16 range: redo exactly
            try:
                # (a lone sample runs the exact kernels at a fraction of their batch throughput: beyond half the batch the whole
                #  pass is the cheaper re-run)
                if call_sample is not None and 0 < len(flagged) <= max(1, batch // 2):
                    for b in flagged:
                        _lib.check(call_sample(b), eng.handle, what + " (exact re-run of sample %d)" % b)
                else:
                    _lib.check(call(eng, 0, batch, C.c_void_p(cur.cuda_stream)), eng.handle, what + " (exact re-run)")
            finally:
                eng.set_precision(self._precision[0])

    def _parts(self, dev, idx, eng, B, allow=True):
        """How a batch of B samples is spread over the device's engines / streams."""
        if not allow or self._nstreams[0] < 2 or B < 4:
            return None
        eng2, _ = self._engine_for(dev, slot=1, same_call=True)
        h1 = B - B // 2
        # NEITHER half on the caller's stream: that is normally the process' default (null) stream, whose launches do not run beside
        # another stream's (measured: the halves serialise, 140 against 130 ms per step; on two side streams they overlap)
        return [(eng, 0, h1, self._side_stream((idx, 0))), (eng2, h1, B, self._side_stream((idx, 1)))]

    def _check_inference(self, reverse=False):
        if self._wants_grad():
            raise NotImplementedError(
                "this call has no backward pass in hcflow_amd (built: SR NLL forward, SR / rescaling sampling path, "
                "rescaling forward). Call under torch.no_grad().")
        if self.training and reverse and self._pending_actnorms():
            raise NotImplementedError(
                "un-initialised ActNorm layers in train() mode on the REVERSE path: the reference would fit them to "
                "the reverse-direction activations (ActNorms.py:78-80), which no shipped configuration does; run one "
                "forward (hr -> z) pass first, load a checkpoint / set .inited = True, or call .eval().")

    # -- ActNorm data-dependent initialisation (ActNorms.py:29-43): fitted by the engine during ONE forward pass
    def _pending_actnorms(self):
        """ActNorms the next train()-mode forward pass has to fit. The reference re-arms ``inited = False`` on every step
        below act_norm_start_step and relies on each layer's ``(bias != 0).any()`` short-circuit (ActNorms.py:33-35,
        HCFlow_SR_model.py:462-465); layers already fitted (or found non-zero, one fused device check) are dropped here
        without an extra statistics pass."""
        pend = [(k, m) for k, m in self.named_modules() if isinstance(m, ActNorm2d) and not m.inited]
        if not pend:
            return pend
        unknown = [(k, m) for k, m in pend if not getattr(m, "_hcf_fitted", False)]
        if unknown and all(m.bias.device.type == "cuda" for _, m in unknown):
            with torch.no_grad():
                nz = (torch.stack([m.bias.detach().abs().max() for _, m in unknown]) > 0).cpu()
            for (k, m), f in zip(unknown, nz.tolist()):
                if f:
                    m._hcf_fitted = True
        out = []
        for k, m in pend:
            if getattr(m, "_hcf_fitted", False):
                m.inited = True
            else:
                out.append((k, m))
        return out

    def _arm_actnorm_init(self, eng):
        """train() mode and ``inited == False`` -> the next forward pass fits bias / logs (eval() mode never
        initialises, ActNorms.py:31-32)."""
        pend = self._pending_actnorms() if self.training else []
        if pend:
            eng.actnorm_init_request([k for k, _ in pend])
        return pend

    def _finish_actnorm_init(self, eng, idx, pend):
        if not pend:
            return
        with torch.no_grad():
            for k, m in pend:
                n = m.bias.numel()
                m.bias.copy_(eng.get_param(k + ".bias", n).view_as(m.bias))
                m.logs.copy_(eng.get_param(k + ".logs", n).view_as(m.logs))
                m.inited = True
                m._hcf_fitted = True
        self._broadcast_actnorms(pend)
        # the engine already holds these values: no repack on the next call (unless the broadcast changed them)
        if not self._dist_on():
            self._engines[idx]["stamp"] = self._stamp_of(self._tensor_list())

    @staticmethod
    def _dist_on():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _broadcast_actnorms(self, pend):
        """Under DDP every rank fits its ActNorms to ITS shard (ActNorms.py:28-44 uses local statistics, so the reference
        starts N ranks from N different parameter sets and DDP never re-synchronises them): rank 0's fit is broadcast, in one
        flat tensor, so all replicas start equal."""
        if not pend or not self._dist_on():
            return
        import torch.distributed as dist
        with torch.no_grad():
            flat = torch.cat([t.detach().reshape(-1) for _, m in pend for t in (m.bias, m.logs)])
            dist.broadcast(flat, src=0)
            off = 0
            for _, m in pend:
                for t in (m.bias, m.logs):
                    n = t.numel()
                    t.copy_(flat[off:off + n].view_as(t))
                    off += n

    @staticmethod
    def _prep(t: torch.Tensor, device) -> torch.Tensor:
        return t.detach().to(device=device, dtype=torch.float32).contiguous()

    @staticmethod
    def _stream(idx):
        return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)

    def _inverse(self, lr, eps_std, eps=None, clamp=True, seed=None, sample_offset=0, cache_cond=False):
        """``sample_offset``: this call is samples [offset, offset + B) of a larger (sharded) batch: the device draws are
        those of the global samples (hcf_inverse_ex). ``cache_cond``: keep / reuse the deepest level's conditional features
        while ``lr`` (same tensor, unchanged) and the parameters stay the same (tau sweeps, repeated sampling)."""
        if self._wants_grad() or (torch.is_grad_enabled() and torch.is_tensor(lr) and lr.requires_grad):
            if self.training and self._pending_actnorms():
                raise NotImplementedError(
                    "un-initialised ActNorm layers in train() mode on the REVERSE path: run one forward (hr -> z) pass "
                    "first, load a checkpoint / set .inited = True, or call .eval().")
            dev = self._device()
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            tau = 0.0 if eps_std is None else float(eps_std)
            lr_t = lr.to(device=dev, dtype=torch.float32).contiguous()          # keeps the autograd link to the caller's lr
            return _SRReverseStep.apply(self, lr_t, tau, seed, bool(clamp), eps, *self._params())
        self._check_inference(reverse=True)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())     # follows torch.manual_seed
        if not self.training and not _capturing():      # (a stale pass is drained with a stream sync: never while capturing)
            try:
                return self._inverse_pass(lr, eps_std, eps, clamp, seed, sample_offset, cache_cond, defer=True)
            except _StaleParameters:
                pass                                               # parameters changed since the last call: the checked path
            except _lib.HcfError:
                if not self.__dict__.pop("_ran_unverified", False):
                    raise
                # an error of a pass that ran on UNVERIFIED packs is not the caller's: the checked path decides
        return self._inverse_pass(lr, eps_std, eps, clamp, seed, sample_offset, cache_cond, defer=False)

    def _inverse_pass(self, lr, eps_std, eps, clamp, seed, sample_offset, cache_cond, defer):
        dev = self._device()
        eng, idx = self._engine_for(dev, defer=defer, twin_hint=self._nstreams[0] >= 2 and torch.is_tensor(lr) and lr.dim() == 4
                                    and int(lr.shape[0]) >= 4)
        lr_in = lr
        lr = self._prep(lr, dev)
        B, c, h, w = lr.shape
        assert c == 3
        cfg = self.cfg
        out = torch.empty(B, 3, h * cfg.scale, w * cfg.scale, device=dev, dtype=torch.float32)
        if B == 0:              # an empty batch samples to an empty batch (every op of the reference's reverse path accepts one)
            return out
        shapes = eps_shapes(cfg, B, h, w)
        keep = []
        keep_at = [None] * len(shapes)
        arr = (C.c_void_p * len(shapes))()
        if eps is not None:
            assert len(eps) == len(shapes)
            for i, (e, s) in enumerate(zip(eps, shapes)):
                if e is None:
                    arr[i] = None
                    continue
                e = self._prep(e, dev)
                assert tuple(e.shape) == tuple(s), (tuple(e.shape), s)
                keep.append(e)
                keep_at[i] = e
                arr[i] = e.data_ptr()
        tau = 0.0 if eps_std is None else float(eps_std)
        flags = 0 if clamp else _lib.FLAG_NO_CLAMP
        if self._range_check[0] == "off":
            flags |= _lib.FLAG_NO_RANGE_CHECK
        parts = self._parts(dev, idx, eng, B)
        if cache_cond:
            # the caller's tensor identity + version stand for its contents; any parameter change drops the key (_engine_for)
            # The keyed tensor is HELD while its key is live: a freed tensor's address is handed to the next same-shape batch
            # by the caching allocator (with _version 0 again), which would otherwise hit the key with other contents.
            # The split layout is part of the key: each engine of a split call keeps ITS half's features, so a call that is split
            # differently from the one that filled the caches (set_streams in between) must refill them.
            key = (lr_in.data_ptr(), lr_in._version, tuple(lr_in.shape), str(lr_in.dtype), str(lr_in.device),
                   0 if parts is None else len(parts))
            have = self._cond_key.get(idx)
            hit = have is not None and have[0] == key and have[1] is lr_in
            flags |= _lib.FLAG_REUSE_COND if hit else _lib.FLAG_KEEP_COND
            self._cond_key[idx] = (key, lr_in)
        else:
            self._cond_key.pop(idx, None)
        def arr_for(lo, hi):
            if lo == 0 and hi == B:
                return arr
            a_ = (C.c_void_p * len(shapes))()
            for i, e in enumerate(keep_at):
                a_[i] = None if e is None else e[lo:hi].data_ptr()     # contiguous NCHW row slices, no copies
            return a_

        def run(eng_, lo, hi, stream_, fl=flags):
            # samples [lo, hi): their LR rows, their rows of the injected draws (or the same seed with their global sample index:
            # the device draws are indexed by sample, hcf_inverse_ex), their output rows
            return eng_.lib.hcf_inverse_ex(eng_.handle, lr[lo:hi].data_ptr(), arr_for(lo, hi), len(shapes), tau, seed,
                                           int(sample_offset) + lo, out[lo:hi].data_ptr(), hi - lo, h, w, fl, stream_)

        def one_sample(b):
            return run(eng, b, b + 1, self._stream(idx), flags & ~(_lib.FLAG_KEEP_COND | _lib.FLAG_REUSE_COND))
        self._run_checked(eng, idx, run, "hcf_inverse", batch=B, call_sample=None if cache_cond else one_sample,
                          parts=parts, threaded=_split_threaded())
        return out

    # convenience for benchmarks / multi-GPU sharding
    def engine(self):
        return self._engine_for(self._device())[0]


class HCFlowNet_SR(_EngineModule):
    """Drop-in for models.modules.HCFlowNet_SR_arch.HCFlowNet_SR (HCFlowNet_SR_arch.py:11-75)."""

    def __init__(self, opt, step=None):
        super(HCFlowNet_SR, self).__init__()
        self.quant = opt_get(opt, ['quant'], 256)
        self._setup(opt)
        assert self.cfg.sr

    # hr: HR image, lr: LR image, z: latent variable, u: conditional variable
    def forward(self, hr=None, lr=None, z=None, u=None, eps_std=None,
                add_gt_noise=False, step=None, reverse=False, training=True, eps=None, noise=None, seed=None,
                sample_offset=0, cache_cond=False):
        if not reverse:
            return self.normal_flow_diracLR(hr, lr, u, step=step, training=training, noise=noise)
        return self.reverse_flow_diracLR(lr, z, u, eps_std=eps_std, training=training, eps=eps, seed=seed,
                                         sample_offset=sample_offset, cache_cond=cache_cond)

    def normal_flow_diracLR(self, hr, lr, u=None, step=None, training=True, noise=None, return_internals=False):
        """hr -> (clamp(LR^), nll)   (HCFlowNet_SR_arch.py:47-67). ``noise``: optional injected U[0,1)
        tensor replacing the internal torch.rand draw (:52). With autograd enabled on trainable parameters the
        pass runs through the engine's taped training path and ``nll.backward()`` works (HCFlow_SR_model.py:195-199)."""
        if self._wants_grad() and not return_internals:
            return self._normal_flow_train(hr, lr, noise)
        self._check_inference()
        dev = self._device()
        eng, idx = self._engine_for(dev)
        self._cond_key.pop(idx, None)
        hr = self._prep(hr, dev)
        B, c, H, W = hr.shape
        s = self.cfg.scale
        assert H % s == 0 and W % s == 0, "{}".format((H, W, 2))
        if noise is None:
            noise = torch.rand(hr.shape, device=dev)
        noise = self._prep(noise, dev)
        lr_t = None if lr is None else self._prep(lr, dev)
        out_lr = torch.empty(B, 3, H // s, W // s, device=dev)
        nll = torch.empty(1, device=dev)
        logdet = torch.empty(B, device=dev)
        zraw = torch.empty(B, 3, H // s, W // s, device=dev) if return_internals else None
        pend = self._arm_actnorm_init(eng)
        self._run_checked(eng, idx, lambda e_, lo, hi, stream: e_.lib.hcf_forward_sr(
            e_.handle, hr.data_ptr(), None if lr_t is None else lr_t.data_ptr(), noise.data_ptr(), out_lr.data_ptr(),
            nll.data_ptr(), logdet.data_ptr(), None if zraw is None else zraw.data_ptr(), B, H, W, stream), "hcf_forward_sr",
            batch=B)                       # (the NLL is a mean over the batch: one pass, one stream)
        self._finish_actnorm_init(eng, idx, pend)
        if return_internals:
            return out_lr, nll[0], logdet, zraw
        return out_lr, nll[0]

    def _normal_flow_train(self, hr, lr, noise):
        dev = self._device()
        assert lr is not None, "the NLL objective needs lr"
        hr, lr = self._prep(hr, dev), self._prep(lr, dev)
        if noise is None:
            noise = torch.rand(hr.shape, device=dev)
        noise = self._prep(noise, dev)
        if self.training and self._pending_actnorms():
            # the reference fits un-initialised ActNorms inside this very forward (ActNorms.py:78-80, no_grad): do
            # that with one statistics pass on the same batch / noise, then run the differentiable pass
            with torch.no_grad():
                self.normal_flow_diracLR(hr, lr, noise=noise)
        if _grad_nodes() == 2:
            params = self._params()
            early_idx = [i for i, k in enumerate(self._spec_keys) if k.startswith("flow.level0_condFlow.")]
            if early_idx and len(early_idx) < len(params):
                eset = set(early_idx)
                st = _SRNLLTwoPhase(self, hr, lr, noise, params, early_idx)
                out_lr, nll, _ = two_phase_apply(st, [params[i] for i in early_idx], [p for i, p in enumerate(params) if i not in eset])
                return out_lr, nll
        out_lr, nll, _ = _SRNLLStep.apply(self, hr, lr, noise, *self._params())
        return out_lr, nll

    def reverse_flow_diracLR(self, lr, z, u, eps_std, training=True, eps=None, clamp=True, seed=None, sample_offset=0,
                             cache_cond=False):
        """lr (+ sampled z) -> clamp(HR)   (HCFlowNet_SR_arch.py:70-75)."""
        return self._inverse(lr, eps_std, eps=eps, clamp=clamp, seed=seed, sample_offset=sample_offset, cache_cond=cache_cond)


class HCFlowNet_Rescaling(_EngineModule):
    """Drop-in for models.modules.HCFlowNet_Rescaling_arch.HCFlowNet_Rescaling (:13-54)."""

    def __init__(self, opt, step=None):
        super(HCFlowNet_Rescaling, self).__init__()
        self.quant = opt_get(opt, ['datasets', 'train', 'quant'], 256)
        self._setup(opt)
        assert not self.cfg.sr

    def forward(self, hr=None, lr=None, z=None, u=None, eps_std=None,
                add_gt_noise=False, step=None, reverse=False, training=True, eps=None, seed=None, sample_offset=0,
                cache_cond=False):
        if not reverse:
            return self.normal_flow_diracLR(hr, lr, u, step=step, training=training)
        return self.reverse_flow_diracLR(lr, z, u, eps_std=eps_std, training=training, eps=eps, seed=seed,
                                         sample_offset=sample_offset, cache_cond=cache_cond)

    def normal_flow_diracLR(self, hr, lr=None, u=None, step=None, training=True, clamp=True):
        """hr -> (clamp(LR^), z1, z2)   (HCFlowNet_Rescaling_arch.py:39-46)."""
        if self._wants_grad():
            dev = self._device()
            if self.training and self._pending_actnorms():
                with torch.no_grad():                       # fit the ActNorms on this batch first (ActNorms.py:78-80)
                    self.normal_flow_diracLR(hr)
            return _RescaleForwardStep.apply(self, self._prep(hr, dev), bool(clamp), *self._params())
        self._check_inference()
        if not self.training and not _capturing():      # (a stale pass is drained with a stream sync: never while capturing)
            try:
                return self._forward_pass(hr, clamp, defer=True)
            except _StaleParameters:
                pass                                               # parameters changed since the last call: the checked path
            except _lib.HcfError:
                if not self.__dict__.pop("_ran_unverified", False):
                    raise
                # an error of a pass that ran on UNVERIFIED packs is not the caller's: the checked path decides
        return self._forward_pass(hr, clamp, defer=False)

    def _forward_pass(self, hr, clamp, defer):
        dev = self._device()
        eng, idx = self._engine_for(dev, defer=defer, twin_hint=self._nstreams[0] >= 2 and torch.is_tensor(hr) and hr.dim() == 4
                                    and int(hr.shape[0]) >= 4 and not self.training)
        self._cond_key.pop(idx, None)
        hr = self._prep(hr, dev)
        B, c, H, W = hr.shape
        assert H % 4 == 0 and W % 4 == 0, "{}".format((H, W, 2))
        cfg = self.cfg
        out_lr = torch.empty(B, 3, H // 4, W // 4, device=dev)
        c0 = cfg.level_channels(0) - cfg.split_channels(0)
        c1 = cfg.level_channels(1) - cfg.split_channels(1)
        z1 = torch.empty(B, c0, H // 2, W // 2, device=dev)
        z2 = torch.empty(B, c1, H // 4, W // 4, device=dev)
        pend = self._arm_actnorm_init(eng)
        fl = (0 if clamp else _lib.FLAG_NO_CLAMP) | (_lib.FLAG_NO_RANGE_CHECK if self._range_check[0] == "off" else 0)
        self._run_checked(eng, idx, lambda e_, lo, hi, stream: e_.lib.hcf_forward_rescale(
            e_.handle, hr[lo:hi].data_ptr(), out_lr[lo:hi].data_ptr(), z1[lo:hi].data_ptr(), z2[lo:hi].data_ptr(), hi - lo, H, W,
            fl, stream), "hcf_forward_rescale", batch=B, parts=self._parts(dev, idx, eng, B, allow=not pend),
            threaded=_split_threaded())
        self._finish_actnorm_init(eng, idx, pend)
        return out_lr, z1, z2

    def reverse_flow_diracLR(self, lr, z, u, eps_std, training=True, eps=None, clamp=True, seed=None, sample_offset=0,
                             cache_cond=False):
        """lr (+ sampled z) -> clamp(HR)   (HCFlowNet_Rescaling_arch.py:49-54)."""
        return self._inverse(lr, eps_std, eps=eps, clamp=clamp, seed=seed, sample_offset=sample_offset, cache_cond=cache_cond)

    def get_score(self, disc_loss_sigma, z):
        """HCFlowNet_Rescaling.get_score (:57-60), unused by every config; kept for API parity."""
        score_real = 0.5 * (1 - 1 / (disc_loss_sigma ** 2)) * (z ** 2).sum(dim=[1, 2, 3]) - \
            z.shape[1] * z.shape[2] * z.shape[3] * math.log(disc_loss_sigma)
        return -score_real
