"""ctypes front end of oracle/hcflow_net.c  --  TEST INFRASTRUCTURE, NOT PRODUCT.

The plain-C restatement of the whole HCFlow forward / inverse path (a second CPU oracle, independent of the PyTorch-CPU one in
oracle/hcflow_oracle.py: no ATen, its own layer plan derived from the state-dict keys inside the C file). Only ``tests/`` and
``bench.py``'s ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libhcflow_net.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_SO)
        L.hcfnet_create.restype = C.c_void_p
        L.hcfnet_free.argtypes = [C.c_void_p]
        L.hcfnet_error.restype = C.c_char_p
        L.hcfnet_error.argtypes = [C.c_void_p]
        L.hcfnet_add_param.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.hcfnet_configure.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float] + [C.c_int] * 8
        L.hcfnet_inverse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int,
                                     C.c_void_p]
        L.hcfnet_sr_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p] * 4
        L.hcfnet_rescale_forward.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p, C.POINTER(C.c_void_p)]
        L.hcfnet_threads.argtypes = [C.c_int]
        L.hcfnet_threads.restype = C.c_int
        _lib = L
    return _lib


def _np(a):
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


class CNet:
    """One network (state dict + the yml values of NetConfig) behind the C restatement."""

    def __init__(self, params: Dict[str, object], cfg):
        L = lib()
        self.cfg = cfg
        self.h = C.c_void_p(L.hcfnet_create())
        self._keep = []
        for k, v in params.items():
            a = _np(v)
            self._keep.append(a)
            dims = (C.c_int * 4)(*(list(a.shape) + [1] * 4)[:4])
            L.hcfnet_add_param(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), min(a.ndim, 4), dims)
        L.hcfnet_configure(self.h, int(cfg.sr), int(cfg.squeeze == "haar"), float(cfg.quant),
                           int(cfg.perm == "invconv"), int(cfg.coupling == "Affine3shift"), int(cfg.nn_module == "DenseBlock"),
                           int(cfg.c_perm == "invconv"), int(cfg.c_coupling == "Affine3shift"),
                           int(cfg.c_nn_module == "DenseBlock"), int(cfg.rrdb_nb[0]), int(cfg.rrdb_nb[1]))

    def __del__(self):
        try:
            lib().hcfnet_free(self.h)
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed: %s" % (what, lib().hcfnet_error(self.h).decode()))

    @staticmethod
    def threads(n: int = 0) -> int:
        return lib().hcfnet_threads(int(n))

    def inverse(self, lr, eps: Optional[Sequence] = None, clamp: bool = True) -> np.ndarray:
        """reverse_flow_diracLR: ``eps`` = the N(0, tau) draws in sampling order, already scaled by tau (None: tau = 0)."""
        lr = _np(lr)
        B, _, h, w = lr.shape
        s = self.cfg.scale
        out = np.empty((B, 3, h * s, w * s), np.float32)
        ea = [_np(e) for e in (eps or [])]
        ptrs = (C.c_void_p * max(1, len(ea)))(*[e.ctypes.data for e in ea])
        self._check(lib().hcfnet_inverse(self.h, lr.ctypes.data, B, h, w, ptrs if ea else None, len(ea), int(clamp),
                                         out.ctypes.data), "hcfnet_inverse")
        return out

    def sr_forward(self, hr, lr, noise):
        """normal_flow_diracLR of the SR net: (clamp(LR^), nll, z before Quant, logdet per sample)."""
        hr, lr, noise = _np(hr), _np(lr), _np(noise)
        B, _, H, W = hr.shape
        lh, lw = lr.shape[2:]
        lr_hat = np.empty_like(lr)
        z = np.empty_like(lr)
        nll = np.zeros(1, np.float32)
        ld = np.zeros(B, np.float32)
        self._check(lib().hcfnet_sr_forward(self.h, hr.ctypes.data, lr.ctypes.data, noise.ctypes.data, B, H, W, lh, lw,
                                            lr_hat.ctypes.data, nll.ctypes.data, z.ctypes.data, ld.ctypes.data), "hcfnet_sr_forward")
        return lr_hat, float(nll[0]), z, ld

    def rescale_forward(self, hr, fake_shapes: List[tuple]):
        """normal_flow_diracLR of the rescaling net: (clamp(LR^), [fake_z per level])."""
        hr = _np(hr)
        B, _, H, W = hr.shape
        s = self.cfg.scale
        lr_hat = np.empty((B, 3, H // s, W // s), np.float32)
        fz = [np.empty(sh, np.float32) for sh in fake_shapes]
        ptrs = (C.c_void_p * len(fz))(*[a.ctypes.data for a in fz])
        self._check(lib().hcfnet_rescale_forward(self.h, hr.ctypes.data, B, H, W, H // s, W // s, lr_hat.ctypes.data, ptrs),
                    "hcfnet_rescale_forward")
        return lr_hat, fz
