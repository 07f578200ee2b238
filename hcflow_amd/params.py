"""Seeded, reference-independent parameter recipe.

No pretrained weights ship with the reference (experiments/pretrained_models/README.md) and a
freshly constructed reference net is a near-identity: ActNorm bias/logs and every Conv2dZeros
are exactly zero (ActNorms.py:19-20, Basic.py:67-68), so it exercises nothing (SURVEY.md
section 0 item 5). Parity and bench runs therefore use this deterministic recipe: every tensor
of ``param_spec(cfg)`` is drawn from one ``torch.Generator`` (CPU, mt19937) in spec order with a
per-kind distribution that keeps the 52-step inverse finite and mostly un-clamped.

The same recipe runs (a) in tests/golden/make_golden.py, where the tensors are loaded into the
*reference* modules with ``load_state_dict(strict=True)`` to produce golden outputs, and (b) on
the GPU box, where the reference is absent, to regenerate identical weights for the engine and
the oracle. ``param_digest`` is stored beside each golden fixture to detect drift.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import NetConfig, param_spec


def _orthogonal(C: int, gen: torch.Generator) -> torch.Tensor:
    """Well-conditioned C x C matrix: Q * diag(exp(N(0, 0.05))) with Q from a float64 QR.

    The reference initialises InvertibleConv1x1 with a random orthogonal matrix
    (Permutations.py:37); trained weights drift away from orthogonality, which the diagonal
    scaling imitates so that slogdet != 0.
    """
    a = torch.randn(C, C, generator=gen, dtype=torch.float64)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r)).unsqueeze(0)     # unique factorisation
    d = torch.exp(0.05 * torch.randn(C, generator=gen, dtype=torch.float64))
    return (q * d.unsqueeze(0)).to(torch.float32).contiguous()


def _lu_factors(C: int, gen: torch.Generator) -> Dict[str, torch.Tensor]:
    """The five tensors of an LU-decomposed InvertibleConv1x1 (Permutations.py:41-57) for a seeded well-conditioned W:
    W = P L (U + diag(sign_s exp(log_s))), partial pivoting in float64 (what scipy.linalg.lu does there), plus a small
    perturbation of l / u / log_s so that the stored factors are not exactly those of an orthogonal matrix."""
    w = _orthogonal(C, gen).double()
    P, L, U = torch.linalg.lu(w)                       # w = P @ L @ U
    s = torch.diagonal(U)
    out = {
        "lu_p": P,
        "lu_sign_s": torch.sign(s),
        "lu_log_s": torch.log(torch.abs(s)) + 0.02 * torch.randn(C, generator=gen, dtype=torch.float64),
        "lu_l": L + torch.tril(0.02 * torch.randn(C, C, generator=gen, dtype=torch.float64), -1),
        "lu_u": torch.triu(U, 1) + torch.triu(0.02 * torch.randn(C, C, generator=gen, dtype=torch.float64), 1),
    }
    return {k: v.to(torch.float32).contiguous() for k, v in out.items()}


def make_params(cfg: NetConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device="cpu")
    gen.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    lu = None
    for key, shape, kind in param_spec(cfg):
        if kind == "conv_w":
            cout, cin, kh, kw = shape
            std = 0.1 * math.sqrt(2.0 / ((cin + cout) * kh * kw)) * 3.0
            t = torch.randn(shape, generator=gen) * std
        elif kind == "conv_b":
            t = torch.randn(shape, generator=gen) * 0.02
        elif kind == "fcn_w":
            cout, cin, kh, kw = shape
            t = torch.randn(shape, generator=gen) * (0.7 / math.sqrt(cin * kh * kw))
        elif kind == "an_bias":
            t = torch.randn(shape, generator=gen) * 0.05
        elif kind == "an_logs":
            t = torch.randn(shape, generator=gen) * 0.05
        elif kind == "zeros_w":
            cout, cin, kh, kw = shape
            t = torch.randn(shape, generator=gen) * (0.08 / math.sqrt(cin * kh * kw))
        elif kind == "zeros_b":
            t = torch.randn(shape, generator=gen) * 0.01
        elif kind == "zeros_logs":
            t = torch.randn(shape, generator=gen) * 0.03
        elif kind == "invconv":
            t = _orthogonal(shape[0], gen)
        elif kind.startswith("lu_"):
            if kind == "lu_l":                       # first of a step's five LU tensors in spec order
                lu = _lu_factors(shape[0], gen)
            t = lu[kind]
        elif kind == "haar":
            # HaarDownsampling.haar_weights (Basic.py:455-468): frozen +-1 pattern
            w = torch.ones(4, 1, 2, 2)
            w[1, 0, 0, 1] = -1
            w[1, 0, 1, 1] = -1
            w[2, 0, 1, 0] = -1
            w[2, 0, 1, 1] = -1
            w[3, 0, 1, 0] = -1
            w[3, 0, 0, 1] = -1
            t = torch.cat([w] * (shape[0] // 4), 0)
        else:
            raise KeyError(kind)
        out[key] = t.to(torch.float32).contiguous()
    return out


def param_digest(params: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Tolerance-comparable fingerprint of a parameter set (float64 statistics)."""
    s1 = 0.0
    s2 = 0.0
    n = 0
    probe = 0.0
    for i, (k, v) in enumerate(sorted(params.items())):
        d = v.double().flatten()
        s1 += float(d.sum())
        s2 += float((d * d).sum())
        n += d.numel()
        probe += float(d[(i * 7919) % d.numel()]) * ((i % 13) + 1)
    return {"n": float(n), "sum": s1, "sumsq": s2, "probe": probe}


def digest_close(a: Dict[str, float], b: Dict[str, float], rtol: float = 1e-6) -> bool:
    for k in ("n", "sum", "sumsq", "probe"):
        if abs(a[k] - b[k]) > rtol * max(1.0, abs(a[k]), abs(b[k])):
            return False
    return True
