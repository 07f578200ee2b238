"""-m gpu: per-op parity of the HIP kernels (through the C ABI) against the CPU oracle.

Tolerances: index ops bit-exact; elementwise 1e-6; one conv 1e-5 relative to the output scale
(fp32 MFMA is an exact fp32 fma chain, only the summation order differs from oneDNN).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import hcflow_oracle as O
from tests.util import load_golden, params_for, t, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from hcflow_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _rel(a, b):
    return maxdiff(a, b) / max(1e-6, float(b.abs().max()))


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ conv
CONV_CASES = [
    # (B, H, W, [src channels], [ups], cout, k, act)
    (2, 8, 32, [16], [0], 32, 3, None),          # exactly one tile, one chunk, one N tile
    (1, 8, 32, [4], [0], 3, 3, None),            # minimal channels: catches M/N/K layout swaps
    (2, 13, 45, [64], [0], 64, 3, "lrelu"),      # ragged edges, RDB conv5 shape
    (1, 20, 20, [64, 96], [0, 0], 32, 3, "lrelu"),   # RDB conv4: x + growth slab
    (2, 10, 12, [10, 128], [0, 0], 64, 3, "relu"),   # cond coupling conv1: cat(z1, u), partial unit
    (1, 16, 24, [6, 128], [0, 1], 64, 3, None),      # level-0 conv_first: cat(z, up2(cf))
    (1, 16, 16, [6, 128, 128], [0, 1, 2], 64, 3, None),  # x8 level-0 conv_first (3 sources)
    (2, 9, 33, [64], [0], 22, 3, None),          # Conv2dZeros head, cout not multiple of 32
    (2, 9, 33, [128], [0], 90, 3, None),         # x8 prior head: 3 N tiles
    (2, 12, 40, [64], [0], 64, 1, "relu"),       # FCN conv2 1x1
    (1, 7, 5, [3], [0], 64, 3, None),            # image smaller than a tile
    (1, 40, 64, [192], [0], 64, 3, None),        # K = 12 chunks, several tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(dev, case):
    from hcflow_amd import ops
    B, H, W, cs, ups, cout, k, act = case
    g = _gen(hash(case[:3]) % 1000 + cout)
    srcs = [torch.randn(B, c, H >> u, W >> u, generator=g) for c, u in zip(cs, ups)]
    cin = sum(cs)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    scale = torch.exp(torch.randn(cout, generator=g) * 0.1)
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(srcs, ups)], 1)
    ref = (F.conv2d(x, w, None, 1, k // 2) + bias.view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    out = ops.conv2d([s.to(dev) for s in srcs], w, bias, scale, act, ups)
    assert _rel(out, ref) <= 1e-5, case
    # plain conv, no epilogue extras
    out2 = ops.conv2d([s.to(dev) for s in srcs], w, None, None, None, ups)
    assert _rel(out2, F.conv2d(x, w, None, 1, k // 2)) <= 1e-5, case


def test_conv2d_residual_epilogues(dev):
    from hcflow_amd import ops
    g = _gen(5)
    x = torch.randn(2, 64, 11, 37, generator=g)
    grow = torch.randn(2, 128, 11, 37, generator=g)
    w = torch.randn(64, 192, 3, 3, generator=g) * 0.02
    b = torch.randn(64, generator=g) * 0.1
    x0 = torch.randn(2, 64, 11, 37, generator=g)
    conv = F.conv2d(torch.cat([x, grow], 1), w, b, 1, 1)
    ref1 = conv * 0.2 + x                       # RDB tail (Basic.py:385)
    ref2 = ref1 * 0.2 + x0                      # RRDB tail (Basic.py:398)
    o1 = ops.conv2d([x.to(dev), grow.to(dev)], w, b, None, None, None, res1=x.to(dev), rs1=0.2)
    o2 = ops.conv2d([x.to(dev), grow.to(dev)], w, b, None, None, None, res1=x.to(dev), rs1=0.2, res2=x0.to(dev), rs2=0.2)
    assert _rel(o1, ref1) <= 1e-5 and _rel(o2, ref2) <= 1e-5


@pytest.mark.precisions("exact")        # f16x3 cannot split a non-finite input: it raises the range flag and the caller re-runs exactly
@pytest.mark.parametrize("act", [None, "relu", "lrelu"])
@pytest.mark.parametrize("cout", [32, 12])          # 16-byte (LDS-transposed) and scalar epilogue
def test_conv2d_non_finite_values_propagate_like_torch(dev, act, cout):
    """NaN / inf pre-activations must come out as torch's conv + activation produce them (the branch-free activation in
    the epilogue may not swallow a NaN; relu(-inf) = 0). Exact fp32-MFMA kernels: the ones a range fallback lands on."""
    from hcflow_amd import ops
    g = _gen(11)
    x = torch.randn(1, 16, 12, 40, generator=g)
    x[0, 3, 5, 7] = float("nan")
    x[0, 2, 9, 30] = float("inf")
    x[0, 1, 2, 20] = float("-inf")
    w = torch.zeros(cout, 16, 3, 3)
    for c in range(cout):
        w[c, c % 16, 1, 1] = 1.0 if c % 2 == 0 else -1.0       # centre tap only: no inf - inf inside the sum
    ref = F.conv2d(x, w, None, 1, 1)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.2)
    out = ops.conv2d([x.to(dev)], w, None, None, act).cpu()
    assert torch.equal(torch.isnan(out), torch.isnan(ref))
    fin = ~torch.isnan(ref)
    assert torch.equal(out[fin], ref[fin])


def test_conv2d_identity_kernel_is_exact(dev, hcf_default_precision):
    """A=I style check with an asymmetric input: centre-tap identity weights must copy x bit-exactly
    (catches transposed fragment layouts that symmetric data would hide). f16x3: x = hi + lo to 2^-22 of each value, with the
    split's absolute floor below |x| = 0.125 (the lo half becomes an f16 subnormal: spacing 2^-24)."""
    from hcflow_amd import ops
    C = 24
    x = torch.arange(2 * C * 9 * 35, dtype=torch.float32).reshape(2, C, 9, 35) * 1e-3
    w = torch.zeros(C, C, 3, 3)
    for c in range(C):
        w[c, c, 1, 1] = 1.0
    out = ops.conv2d([x.to(dev)], w)

    def same(a, b):
        if hcf_default_precision == "exact":
            return torch.equal(a, b)
        return bool(((a - b).abs() <= 2.0 ** -21 * b.abs() + 2.0 ** -24).all())
    assert same(out.cpu(), x)
    # shifted tap: out[y, x] = in[y, x+1]  (tap kx = 2), zero padded at the right edge
    w2 = torch.zeros(C, C, 3, 3)
    for c in range(C):
        w2[c, (c + 1) % C, 1, 2] = 1.0
    ref = F.conv2d(x, w2, None, 1, 1)
    assert same(ops.conv2d([x.to(dev)], w2).cpu(), ref)


# ------------------------------------------------------------------ index ops
def test_squeeze_unsqueeze_bit_exact(dev):
    from hcflow_amd import ops
    g = load_golden("ops_index")
    assert torch.equal(ops.squeeze2d(t(g["sq_in"]).to(dev)).cpu(), t(g["sq_out"]))
    assert torch.equal(ops.unsqueeze2d(t(g["usq_in"]).to(dev)).cpu(), t(g["usq_out"]))
    x = torch.randn(3, 6, 14, 10, generator=_gen(1))
    assert torch.equal(ops.unsqueeze2d(ops.squeeze2d(x.to(dev))).cpu(), x)
    assert torch.equal(ops.squeeze2d(x.to(dev)).cpu(), O.squeeze2d(x))


def test_haar(dev):
    from hcflow_amd import ops
    g = load_golden("ops_index")
    assert maxdiff(ops.squeeze2d(t(g["haar_in"]).to(dev), haar=True), g["haar_fwd"]) <= 1e-6
    assert maxdiff(ops.unsqueeze2d(t(g["haar_inv_in"]).to(dev), haar=True), g["haar_inv_out"]) <= 1e-6
    x = torch.randn(2, 3, 12, 8, generator=_gen(2))
    assert maxdiff(ops.unsqueeze2d(ops.squeeze2d(x.to(dev), haar=True), haar=True), x) <= 1e-6


# ------------------------------------------------------------------ flow step glue
@pytest.mark.parametrize("C,ns,hw", [(12, 6, (6, 10)), (24, 12, (5, 7)), (21, 10, (5, 7)), (6, 3, (9, 4)), (45, 22, (3, 5)), (48, 24, (4, 4))])
def test_step_affine(dev, C, ns, hw):
    from hcflow_amd import ops
    g = _gen(C)
    H, W = hw
    z = torch.randn(2, C, H, W, generator=g)
    h = torch.randn(2, 2 * (C - ns), H, W, generator=g) * 0.5
    Wm = torch.linalg.qr(torch.randn(C, C, generator=g, dtype=torch.float64))[0].float() * 1.1
    bias = torch.randn(1, C, 1, 1, generator=g) * 0.1
    logs = torch.randn(1, C, 1, 1, generator=g) * 0.1
    # oracle pieces (AffineCouplings.py:65-87, Permutations.py:72-74, ActNorms.py:54,66)
    shift, scale = O.split_cross(h)
    z2 = z[:, ns:] * torch.exp(-O.logscale_of(scale)) - shift
    ref = O.actnorm_inverse(O.invconv_inverse(torch.cat((z[:, :ns], z2), 1), Wm), bias, logs)
    out = ops.step_inverse(z.to(dev), h.to(dev), 0, ns, Wm, bias, logs)
    assert maxdiff(out, ref) <= 2e-6 * max(1.0, float(ref.abs().max()))
    # forward head + couple
    mid = O.invconv_forward(O.actnorm_forward(z, bias, logs), Wm)
    o_mid = ops.step_forward_head(z.to(dev), Wm, bias, logs)
    assert maxdiff(o_mid, mid) <= 2e-6 * max(1.0, float(mid.abs().max()))
    ls = O.logscale_of(scale)
    fwd = torch.cat((mid[:, :ns], (mid[:, ns:] + shift) * torch.exp(ls)), 1)
    o_fwd, ld = ops.step_forward_couple(mid.to(dev), h.to(dev), 0, ns)
    assert maxdiff(o_fwd, fwd) <= 2e-6 * max(1.0, float(fwd.abs().max()))
    assert maxdiff(ld, O.sum_chw(ls)) <= 1e-4


def test_step_shift3_and_no_perm(dev):
    from hcflow_amd import ops
    g = _gen(3)
    z = torch.randn(2, 12, 6, 10, generator=g)
    h = torch.randn(2, 3, 6, 10, generator=g)
    bias = torch.randn(1, 12, 1, 1, generator=g) * 0.1
    logs = torch.randn(1, 12, 1, 1, generator=g) * 0.1
    ref = O.actnorm_inverse(torch.cat((z[:, :3] - h, z[:, 3:]), 1), bias, logs)   # AffineCouplings.py:150-158
    out = ops.step_inverse(z.to(dev), h.to(dev), 1, 3, None, bias, logs)
    assert maxdiff(out, ref) <= 1e-6
    o_fwd, ld = ops.step_forward_couple(z.to(dev), h.to(dev), 1, 3)
    assert maxdiff(o_fwd, torch.cat((z[:, :3] + h, z[:, 3:]), 1)) <= 1e-6
    assert float(ld.abs().max()) == 0.0


def test_flowstep_against_golden(dev):
    """Full FlowStep inverse via conv + tail ops on the reference-generated fixture."""
    from hcflow_amd import ops
    g = load_golden("ops_sr_tiny")
    cfg, p = params_for(g)
    pre = "flow.layers.1"
    z = t(g["fs_fwd"]).to(dev)
    f = pre + ".affine.f"
    h1 = ops.conv2d([z[:, :6]], p[f + ".conv1.weight"], p[f + ".conv1.actnorm.bias"].flatten(),
                    torch.exp(p[f + ".conv1.actnorm.logs"].flatten()), "relu")
    h2 = ops.conv2d([h1], p[f + ".conv2.weight"], p[f + ".conv2.actnorm.bias"].flatten(),
                    torch.exp(p[f + ".conv2.actnorm.logs"].flatten()), "relu")
    h = ops.conv2d([h2], p[f + ".conv3.weight"], p[f + ".conv3.bias"], torch.exp(p[f + ".conv3.logs"].flatten() * 3))
    out = ops.step_inverse(z, h, 0, 6, p[pre + ".permute.weight"], p[pre + ".actnorm.bias"], p[pre + ".actnorm.logs"])
    assert maxdiff(out, g["fs_inv_of_fwd"]) <= 1e-5
    assert maxdiff(out, g["fs_in"]) <= 1e-4


# ------------------------------------------------------------------ gaussian prior
def test_gauss(dev):
    from hcflow_amd import ops
    g = load_golden("ops_index")
    mean, logs, x = t(g["g_mean"]), t(g["g_logs"]), t(g["g_x"])
    h = torch.stack((mean, logs), 2).reshape(2, 12, 4, 4)          # "cross" interleave
    lp = ops.gauss_logp(h.to(dev), x.to(dev))
    assert maxdiff(lp, g["g_logp"]) <= 1e-4 * max(1.0, float(np.abs(g["g_logp"]).max()))
    eps = torch.randn(2, 6, 4, 4, generator=_gen(9)) * 0.8
    s = ops.gauss_sample(h.to(dev), eps.to(dev))
    assert maxdiff(s, mean + torch.exp(logs) * eps) <= 1e-6
    s0 = ops.gauss_sample(h.to(dev), None, tau=0.0)                # torch.normal(std=0) == mean exactly
    assert torch.equal(s0.cpu(), mean)
    sr = ops.gauss_sample(h.to(dev), eps.to(dev), rescale=True)
    assert maxdiff(sr, mean + torch.exp(O.logscale_of(logs)) * eps) <= 1e-6


def test_device_sampler_statistics(dev):
    from hcflow_amd import ops
    h = torch.zeros(4, 2 * 8, 64, 64, device=dev)                 # mean 0, logs 0
    a = ops.gauss_sample(h, None, tau=0.8, seed=123)
    b = ops.gauss_sample(h, None, tau=0.8, seed=123)
    c = ops.gauss_sample(h, None, tau=0.8, seed=124)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.mean())) < 0.01 and abs(float(a.std()) - 0.8) < 0.01
    k = float(((a / 0.8) ** 4).mean())
    assert abs(k - 3.0) < 0.15                                     # Gaussian kurtosis
