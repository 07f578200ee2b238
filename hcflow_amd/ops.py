"""Thin torch wrappers over the per-op C-ABI entry points (include/hcflow.h, ``hcf_op_*``).

Used by the unit parity tests; every function takes CUDA fp32 NCHW tensors and returns a new CUDA
tensor computed by the HIP kernels. No fallback paths.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib


def _dev(t: torch.Tensor) -> torch.Tensor:
    if t.device.type != "cuda":
        raise _lib.HcfError("hcflow_amd ops need CUDA (MI355X) tensors")
    return t.detach().to(torch.float32).contiguous()


def _host(t: Optional[torch.Tensor]):
    if t is None:
        return None, None
    h = t.detach().to("cpu", torch.float32).contiguous()
    return h, C.c_void_p(h.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


ACT = {None: 0, "none": 0, "relu": 1, "lrelu": 2}


def set_precision(mode: str):
    """Numerics of conv2d() below: "exact" (fp32 MFMA) or "f16x3" (split products on f16 MFMA)."""
    _lib.check(_lib.load().hcf_op_set_precision(_lib.Engine.PRECISIONS[mode]), None, "hcf_op_set_precision")


def conv2d(srcs: Sequence[torch.Tensor], weight: torch.Tensor, bias=None, scale=None, act=None,
           ups: Optional[Sequence[int]] = None, res1=None, rs1=0.0, res2=None, rs2=0.0) -> torch.Tensor:
    """act((conv(cat(upsampled srcs), weight) + bias) * scale) [* rs1 + res1] [* rs2 + res2]."""
    lib = _lib.load()
    srcs = [_dev(s) for s in srcs]
    n = len(srcs)
    ups = list(ups) if ups is not None else [0] * n
    B = srcs[0].shape[0]
    H, W = srcs[0].shape[2] << ups[0], srcs[0].shape[3] << ups[0]
    cout, cin, k, _ = weight.shape
    assert sum(s.shape[1] for s in srcs) == cin
    ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    cs = (C.c_int32 * n)(*[s.shape[1] for s in srcs])
    us = (C.c_int32 * n)(*ups)
    wh, wp = _host(weight)
    bh, bp = _host(bias)
    sh, sp = _host(scale)
    out = torch.empty(B, cout, H, W, device=srcs[0].device)
    r1 = _dev(res1) if res1 is not None else None
    r2 = _dev(res2) if res2 is not None else None
    rc = lib.hcf_op_conv2d(ptrs, cs, us, n, B, H, W, wp, bp, sp, cout, k, ACT[act],
                           None if r1 is None else r1.data_ptr(), float(rs1),
                           None if r2 is None else r2.data_ptr(), float(rs2), out.data_ptr(), _stream(out))
    _lib.check(rc, None, "hcf_op_conv2d")
    return out


def conv2d_backward(srcs: Sequence[torch.Tensor], weight: torch.Tensor, grad_out: torch.Tensor,
                    ups: Optional[Sequence[int]] = None, need_input_grads: bool = True):
    """torch.autograd of ``F.conv2d(cat(upsampled srcs), weight, bias, 1, k // 2)``: returns
    ([d srcs] or None, d weight (CPU), d bias (CPU))."""
    lib = _lib.load()
    srcs = [_dev(s) for s in srcs]
    g = _dev(grad_out)
    n = len(srcs)
    ups = list(ups) if ups is not None else [0] * n
    B, cout, H, W = g.shape
    co, cin, k, _ = weight.shape
    assert co == cout and sum(s.shape[1] for s in srcs) == cin
    ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    cs = (C.c_int32 * n)(*[s.shape[1] for s in srcs])
    us = (C.c_int32 * n)(*ups)
    wh, wp = _host(weight)
    dsrcs = [torch.empty_like(s) for s in srcs] if need_input_grads else None
    dptrs = (C.c_void_p * n)(*([d.data_ptr() for d in dsrcs] if dsrcs else [None] * n))
    dw = torch.empty(cout, cin, k, k, dtype=torch.float32)
    db = torch.empty(cout, dtype=torch.float32)
    rc = lib.hcf_op_conv2d_backward(ptrs, cs, us, n, B, H, W, wp, cout, k, g.data_ptr(), dptrs,
                                    C.c_void_p(dw.data_ptr()), C.c_void_p(db.data_ptr()), _stream(g))
    _lib.check(rc, None, "hcf_op_conv2d_backward")
    return dsrcs, dw, db


def squeeze2d(x: torch.Tensor, haar: bool = False) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x)
    B, Cc, H, W = x.shape
    out = torch.empty(B, 4 * Cc, H // 2, W // 2, device=x.device)
    _lib.check(lib.hcf_op_squeeze2d(x.data_ptr(), out.data_ptr(), B, Cc, H, W, int(haar), _stream(x)), None,
               "hcf_op_squeeze2d")
    return out


def unsqueeze2d(x: torch.Tensor, haar: bool = False) -> torch.Tensor:
    lib = _lib.load()
    x = _dev(x)
    B, C4, H, W = x.shape
    out = torch.empty(B, C4 // 4, H * 2, W * 2, device=x.device)
    _lib.check(lib.hcf_op_unsqueeze2d(x.data_ptr(), out.data_ptr(), B, C4, H, W, int(haar), _stream(x)), None,
               "hcf_op_unsqueeze2d")
    return out


def step_inverse(z, h, mode: int, ns: int, mat, an_bias, an_logs) -> torch.Tensor:
    lib = _lib.load()
    z, h = _dev(z), _dev(h)
    B, Cc, H, W = z.shape
    out = torch.empty_like(z)
    mh, mp = _host(mat)
    bh, bp = _host(an_bias.flatten())
    lh, lp = _host(an_logs.flatten())
    _lib.check(lib.hcf_op_step_inverse(z.data_ptr(), h.data_ptr(), out.data_ptr(), B, Cc, H, W, h.shape[1], mode, ns,
                                       mp, bp, lp, _stream(z)), None, "hcf_op_step_inverse")
    return out


def step_forward_head(z, mat, an_bias, an_logs) -> torch.Tensor:
    lib = _lib.load()
    z = _dev(z)
    B, Cc, H, W = z.shape
    out = torch.empty_like(z)
    mh, mp = _host(mat)
    bh, bp = _host(an_bias.flatten())
    lh, lp = _host(an_logs.flatten())
    _lib.check(lib.hcf_op_step_forward_head(z.data_ptr(), out.data_ptr(), B, Cc, H, W, mp, bp, lp, _stream(z)), None,
               "hcf_op_step_forward_head")
    return out


def step_forward_couple(z, h, mode: int, ns: int):
    lib = _lib.load()
    z, h = _dev(z), _dev(h)
    B, Cc, H, W = z.shape
    out = torch.empty_like(z)
    ld = torch.empty(B, device=z.device)
    _lib.check(lib.hcf_op_step_forward_couple(z.data_ptr(), h.data_ptr(), out.data_ptr(), ld.data_ptr(), B, Cc, H, W,
                                              h.shape[1], mode, ns, _stream(z)), None, "hcf_op_step_forward_couple")
    return out, ld


def gauss_logp(h, x) -> torch.Tensor:
    lib = _lib.load()
    h, x = _dev(h), _dev(x)
    B, Cc, H, W = x.shape
    out = torch.empty(B, device=x.device)
    _lib.check(lib.hcf_op_gauss_logp(h.data_ptr(), x.data_ptr(), out.data_ptr(), B, Cc, H, W, _stream(x)), None,
               "hcf_op_gauss_logp")
    return out


def gauss_sample(h, eps=None, tau: float = 1.0, seed: int = 0, rescale: bool = False) -> torch.Tensor:
    lib = _lib.load()
    h = _dev(h)
    B, C2, H, W = h.shape
    out = torch.empty(B, C2 // 2, H, W, device=h.device)
    e = _dev(eps) if eps is not None else None
    _lib.check(lib.hcf_op_gauss_sample(h.data_ptr(), None if e is None else e.data_ptr(), float(tau), int(seed),
                                       out.data_ptr(), B, C2 // 2, H, W, int(rescale), _stream(h)), None,
               "hcf_op_gauss_sample")
    return out
