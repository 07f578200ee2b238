"""The plain-C oracle (oracle/hcflow_ref.c) against the reference-generated golden fixtures and the
torch oracle. Small sizes only (scalar loops)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hcflow_oracle as O
from tests.util import load_golden, params_for, t, maxdiff

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fp = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def ref():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return C.CDLL(os.path.join(ROOT, "oracle", "_build", "libhcflow_ref.so"))


def P(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(fp)


def test_c_index_ops(ref):
    g = load_golden("ops_index")
    x, xp = P(g["sq_in"])
    out = np.empty_like(g["sq_out"])
    ref.ref_squeeze2d(xp, out.ctypes.data_as(fp), 2, 3, 8, 12)
    assert np.array_equal(out, g["sq_out"])
    y, yp = P(g["usq_in"])
    out = np.empty_like(g["usq_out"])
    ref.ref_unsqueeze2d(yp, out.ctypes.data_as(fp), 2, 12, 4, 6)
    assert np.array_equal(out, g["usq_out"])
    h, hp = P(g["haar_in"])
    out = np.empty_like(g["haar_fwd"])
    ref.ref_haar_forward(hp, out.ctypes.data_as(fp), 2, 3, 8, 12)
    assert np.abs(out - g["haar_fwd"]).max() <= 1e-6
    h2, h2p = P(g["haar_inv_in"])
    out = np.empty_like(g["haar_inv_out"])
    ref.ref_haar_inverse(h2p, out.ctypes.data_as(fp), 2, 12, 4, 6)
    assert np.abs(out - g["haar_inv_out"]).max() <= 1e-6
    q, qp = P(g["q_in"])
    out = np.empty_like(g["q_out"])
    ref.ref_quant.argtypes = [fp, fp, C.c_size_t]
    ref.ref_quant(qp, out.ctypes.data_as(fp), q.size)
    assert np.array_equal(out, g["q_out"])
    m, mp_ = P(g["g_mean"]); l, lp = P(g["g_logs"]); xx, xxp = P(g["g_x"])
    out = np.empty(2, dtype=np.float32)
    ref.ref_gauss_logp(mp_, lp, xxp, out.ctypes.data_as(fp), 2, 6 * 4 * 4)
    assert np.abs(out - g["g_logp"]).max() <= 1e-4 * max(1.0, np.abs(g["g_logp"]).max())


def test_c_conv_matches_torch(ref):
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 7, 9, generator=gen)
    for k in (1, 3):
        w = torch.randn(4, 5, k, k, generator=gen) * 0.3
        b = torch.randn(4, generator=gen)
        out = np.empty((2, 4, 7, 9), dtype=np.float32)
        xa, xp = P(x.numpy()); wa, wp = P(w.numpy()); ba, bp = P(b.numpy())
        ref.ref_conv2d(xp, wp, bp, out.ctypes.data_as(fp), 2, 5, 7, 9, 4, k)
        assert maxdiff(out, F.conv2d(x, w, b, 1, k // 2)) <= 1e-5


def test_c_flowstep_inverse_matches_reference_fixture(ref):
    g = load_golden("ops_sr_tiny")
    cfg, p = params_for(g)
    for pre, zin, u, want, C_, hw in (
            ("flow.layers.1", g["fs_fwd"], None, g["fs_inv_of_fwd"], 12, (6, 10)),
            ("flow.level1_condFlow.additional_flow_steps.0", g["cs_fwd"], g["cs_u"], g["cs_inv_of_fwd"], 21, (5, 7))):
        f = pre + ".affine.f"
        arrs = [P(p[k].numpy()) for k in (
            pre + ".actnorm.bias", pre + ".actnorm.logs", pre + ".permute.weight",
            f + ".conv1.weight", f + ".conv1.actnorm.bias", f + ".conv1.actnorm.logs",
            f + ".conv2.weight", f + ".conv2.actnorm.bias", f + ".conv2.actnorm.logs",
            f + ".conv3.weight", f + ".conv3.bias", f + ".conv3.logs")]
        z, zp = P(zin)
        if u is not None:
            ua, up = P(u)
        out = np.empty_like(want)
        rc = ref.ref_flowstep_inverse(zp, up if u is not None else None, 128 if u is not None else 0,
                                      out.ctypes.data_as(fp), 2, C_, hw[0], hw[1], 64, *[a[1] for a in arrs])
        assert rc == 0
        assert np.abs(out - want).max() <= 2e-5, pre
        # and against the torch oracle
        o2 = O.flowstep_inverse(t(zin), None if u is None else t(u), p, pre, "invconv", "Affine", "FCN")
        assert maxdiff(out, o2) <= 2e-5


def test_c_affine_forward_logdet(ref):
    gen = torch.Generator().manual_seed(1)
    z = torch.randn(2, 12, 4, 5, generator=gen)
    h = torch.randn(2, 12, 4, 5, generator=gen) * 0.5
    za, zp = P(z.numpy()); ha, hp = P(h.numpy())
    out = np.empty((2, 12, 4, 5), dtype=np.float32)
    ld = np.empty(2, dtype=np.float32)
    ref.ref_affine_coupling(zp, hp, out.ctypes.data_as(fp), ld.ctypes.data_as(fp), 2, 12, 6, 20, 0)
    shift, scale = O.split_cross(h)
    ls = O.logscale_of(scale)
    want = torch.cat((z[:, :6], (z[:, 6:] + shift) * torch.exp(ls)), 1)
    assert maxdiff(out, want) <= 1e-6 and maxdiff(ld, O.sum_chw(ls)) <= 1e-4
