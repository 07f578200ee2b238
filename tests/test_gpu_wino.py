"""-m gpu: the Winograd F(2x2, 3x3) form of the f16x3 convolution (hcf_conv_wino.h / hcf_conv_wino.hip), which the engine
and the op entry use for plain 3x3 convs with >= 64 input channels in 16-channel-aligned source windows and 32 / 64 output
channels (the convs of the residual dense blocks and the 64 -> 64 trunk convs, RRDBNet_arch.py:18-34, 84-97): fp32-class accuracy against an fp64
evaluation, agreement with the direct f16x3 kernel, range reporting, and that the engine really takes it."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_ops import _rel, _gen

pytestmark = pytest.mark.gpu

# B, H, W, source channels, cout, act, residuals
WINO_CASES = [
    (2, 16, 32, [64, 64], 32, "lrelu", 0),          # RDB conv3, exact unit grid
    (1, 24, 40, [64, 96], 32, "lrelu", 0),          # RDB conv4, ragged unit edges (24 = 16 + 8 rows, 40 = 32 + 8 columns)
    (2, 19, 37, [64, 128], 64, None, 1),            # RDB conv5 with the dense-block residual, odd sizes
    (1, 33, 70, [64, 128], 64, None, 2),            # ... and with the RRDB skip as second residual
    (1, 8, 8, [128], 32, "relu", 0),                # a single unit, one source
    (3, 48, 64, [160], 32, "lrelu", 0),             # several units per block
    (2, 20, 36, [64], 32, "lrelu", 0),              # RDB conv1 (the smallest eligible K: 4 chunks)
    (1, 16, 32, [64, 32], 32, "lrelu", 0),          # RDB conv2: a 32-channel second source
    (2, 24, 32, [64], 64, None, 0),                 # trunk conv 64 -> 64 (v4 with 4 chunks)
]


def _ablate(bits):
    from hcflow_amd import _lib
    assert _lib.load().hcf_debug_set_ablation(bits) == 0


@pytest.fixture()
def f16x3_ops():
    from hcflow_amd import ops
    ops.set_precision("f16x3")
    yield ops
    ops.set_precision("exact")
    _ablate(0)


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv2d_winograd_matches_fp64_and_direct(f16x3_ops, case):
    ops = f16x3_ops
    B, H, W, cs, cout, act, nres = case
    g = _gen(sum(cs) + H + W)
    srcs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    cin = sum(cs)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    scale = torch.exp(torch.randn(cout, generator=g) * 0.1)
    res = [torch.randn(B, cout, H, W, generator=g) for _ in range(nres)]
    ref = (F.conv2d(torch.cat(srcs, 1).double(), w.double(), None, 1, 1) + bias.view(1, -1, 1, 1)) * scale.view(1, -1, 1, 1)
    ref = F.relu(ref) if act == "relu" else F.leaky_relu(ref, 0.2) if act == "lrelu" else ref
    kw = {}
    if nres >= 1:
        ref = ref * 0.2 + res[0].double()
        kw.update(res1=res[0].cuda(), rs1=0.2)
    if nres == 2:
        ref = ref * 0.2 + res[1].double()
        kw.update(res2=res[1].cuda(), rs2=0.2)
    dev = [s.cuda() for s in srcs]
    _ablate(0)
    out = ops.conv2d(dev, w, bias, scale, act, **kw)
    out2 = ops.conv2d(dev, w, bias, scale, act, **kw)
    _ablate(256)                                           # the direct f16x3 kernel on the same call
    direct = ops.conv2d(dev, w, bias, scale, act, **kw)
    _ablate(0)
    assert torch.equal(out, out2)
    assert _rel(out, ref.float()) <= 3e-6, (case, _rel(out, ref.float()))
    assert _rel(direct, ref.float()) <= 3e-6
    assert not torch.equal(out, direct)                    # (different summation: proves the other kernel ran)
    assert _rel(out, direct) <= 4e-6


def test_winograd_out_of_range_is_reported(f16x3_ops):
    from hcflow_amd import _lib
    x = torch.ones(1, 128, 16, 32)
    x[0, 77, 9, 21] = 1.0e5                    # beyond the f16 range: the transformed value overflows the hi part
    w = torch.randn(32, 128, 3, 3, generator=_gen(1)) * 0.05
    with pytest.raises(_lib.HcfError):
        f16x3_ops.conv2d([x.cuda()], w)


def test_engine_uses_winograd_for_the_deep_dense_block_convs():
    """SR x4 tiny net, f16x3 inverse pass: conv3 / conv4 / conv5 of every RDB run as kind 4 (the others as before), the result
    stays within the f16x3 tolerance of the exact kernels and is bit-reproducible; --ablate 256 gives the direct kernels."""
    from hcflow_amd.config import preset, eps_shapes
    from tests.util import cached_params, maxdiff
    from tests.test_gpu_nets import build_net
    cfg = preset("SR_4X_tiny")
    net = build_net(cfg, cached_params("SR_4X_tiny", 11)).set_precision("exact")      # the comparisons below start from the exact kernels
    net.invalidate()          # (a cached net that has been through a device-side refresh keeps the per-conv schedule until re-finalised)
    g = torch.Generator().manual_seed(5)
    B, size = 2, 40
    lr = torch.rand(B, 3, size, size, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, size, size)]
    with torch.no_grad():
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        net.set_precision("f16x3")
        try:
            a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            eng = net.engine()
            eng.profile_convs(True)
            b = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            n1 = eng.conv_time(9, 1, kind=4)[1] + eng.conv_time(9, 1, kind=7)[1]     # (7: the completions of the fat launches)
            ms2, n2, _, _ = eng.conv_time(9, 2, kind=4, reset=True)
            _ablate(256)
            d = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            n3 = eng.conv_time(9, 0, kind=4)[1] + eng.conv_time(9, 0, kind=7, reset=True)[1]
            eng.profile_convs(False)
        finally:
            _ablate(0)
            net.set_precision("exact")
    assert n1 > 0 and n2 > 0 and n3 == 0, (n1, n2, n3)
    assert torch.equal(a, b)
    tol = 2e-5 * max(1.0, float(ex.abs().max()))
    assert maxdiff(a, ex) <= tol and maxdiff(d, ex) <= tol


def test_engine_runs_conditional_fcn_conv1_conv2_on_the_winograd_kernel():
    """Conditional coupling nets (Basic.py:441-447 with cat(z1, features), AffineCouplings.py:65-87): conv1 over [z1 padded to
    16 | 128 features] runs as ONE 64-channel Winograd launch with the 1x1 conv2 in its epilogue (kind 6); --ablate 512 gives
    the direct fused kernel (kind 1) back; both stay within the f16x3 tolerance of the exact kernels, and of each other."""
    from hcflow_amd.config import preset, eps_shapes
    from tests.util import cached_params, maxdiff
    from tests.test_gpu_nets import build_net
    cfg = preset("SR_4X_tiny")
    net = build_net(cfg, cached_params("SR_4X_tiny", 11)).set_precision("exact")      # the comparisons below start from the exact kernels
    net.invalidate()          # (a cached net that has been through a device-side refresh keeps the per-conv schedule until re-finalised)
    g = torch.Generator().manual_seed(6)
    B, h, w = 2, 37, 50                       # ragged against the 8 x 32 units
    lr = torch.rand(B, 3, h, w, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, h, w)]
    with torch.no_grad():
        ex = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
        net.set_precision("f16x3")
        try:
            eng = net.engine()
            eng.profile_convs(True)
            a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            _, n6, _, _ = eng.conv_time(9, 2, kind=6)
            _, n1, _, _ = eng.conv_time(9, 2, kind=1, reset=True)
            _ablate(512)
            d = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            _, m6, _, _ = eng.conv_time(9, 2, kind=6)
            _, m1, _, _ = eng.conv_time(9, 2, kind=1, reset=True)
            eng.profile_convs(False)
            assert eng.fallback_count() == 0
        finally:
            _ablate(0)
            net.set_precision("exact")
    n_cond = sum(cfg.after)                   # conditional steps per pass
    assert (n6, n1, m6, m1) == (n_cond, 0, 0, n_cond), (n6, n1, m6, m1, n_cond)
    tol = 2e-5 * max(1.0, float(ex.abs().max()))
    assert maxdiff(a, ex) <= tol and maxdiff(d, ex) <= tol and maxdiff(a, d) <= tol


def test_engine_runs_denseblock_coupling_convs_on_the_winograd_kernels():
    """Rescaling nets: the DenseBlock coupling nets (Basic.py:329-356) read cat(z1, growth): with z1 (3 / 9 / 21 channels) padded to
    whole 16-channel chunks conv2 .. conv5 take the Winograd form (the last one as a zero-padded output tile: 3 / 18 / 42 real
    channels); --ablate 1024 gives the direct kernels back; forward (encode) and inverse (decode) both, within the f16x3 tolerance
    of the exact kernels and of each other."""
    from hcflow_amd.config import preset
    from tests.util import cached_params, maxdiff
    from tests.test_gpu_nets import build_net
    cfg = preset("Rescaling_4X_tiny")
    net = build_net(cfg, cached_params("Rescaling_4X_tiny", 13)).set_precision("exact")
    net.invalidate()
    g = torch.Generator().manual_seed(8)
    hr = torch.rand(2, 3, 136, 200, generator=g).cuda()                 # ragged against the 16 x 32 units at both levels
    n_dense = sum(cfg.K[:cfg.L]) - sum(cfg.after)                       # DenseBlock steps per pass
    with torch.no_grad():
        lr_e = net(hr=hr, reverse=False)[0]
        rec_e = net(lr=lr_e, reverse=True, eps_std=0.0)
        net.set_precision("f16x3")
        try:
            eng = net.engine()
            eng.profile_convs(True)
            lr_a = net(hr=hr, reverse=False)[0]
            n_fwd = eng.conv_time(9, 0, kind=4, reset=True)[1]
            rec_a = net(lr=lr_e, reverse=True, eps_std=0.0)
            n_inv = eng.conv_time(9, 0, kind=4, reset=True)[1]
            _ablate(1024)
            lr_d = net(hr=hr, reverse=False)[0]
            m_fwd = eng.conv_time(9, 0, kind=4, reset=True)[1]
            rec_d = net(lr=lr_e, reverse=True, eps_std=0.0)
            eng.profile_convs(False)
            assert eng.fallback_count() == 0
        finally:
            _ablate(0)
            net.set_precision("exact")
    assert n_fwd - m_fwd == 4 * n_dense and n_inv >= 4 * n_dense, (n_fwd, m_fwd, n_inv, n_dense)
    tol = 2e-5 * max(1.0, float(rec_e.abs().max()))
    assert maxdiff(rec_a, rec_e) <= tol and maxdiff(rec_d, rec_e) <= tol and maxdiff(rec_a, rec_d) <= tol
    # the quantised LR image: a level may flip where the pre-quantisation value sits on a rounding boundary
    for x in (lr_a, lr_d):
        assert float(((x - lr_e).abs() > 0.5 / 255).float().mean()) <= 1e-3


def test_engine_runs_16_channel_dense_blocks_as_fat_pairs():
    """Rescaling nets (RRDB_gc = 16, Basic.py:360-377): four growth convs that each fill HALF of a 32-wide MFMA tile become two fat
    pairs -- conv 2j+1 + the old-input part of conv 2j+2 as ONE 32-channel Winograd launch whose upper half-tile is stored raw
    (Args::out2_split = 16), and a 16 -> 16 completion (one K chunk, zero-padded tile) that adds it before the activation (kind 7).
    Launch counts, and the passes stay within the f16x3 tolerance of the exact kernels."""
    from hcflow_amd.config import preset
    from tests.util import cached_params, maxdiff
    from tests.test_gpu_nets import build_net
    cfg = preset("Rescaling_4X_tiny")
    assert cfg.rrdb_gc == 16
    net = build_net(cfg, cached_params("Rescaling_4X_tiny", 13)).set_precision("exact")
    net.invalidate()
    g = torch.Generator().manual_seed(9)
    hr = torch.rand(2, 3, 136, 200, generator=g).cuda()                 # ragged against the 16 x 32 units at both levels
    n_rdb = 3 * (cfg.rrdb_nb[0] + cfg.rrdb_nb[1]) * cfg.L               # dense blocks per pass (both levels)
    with torch.no_grad():
        lr_e = net(hr=hr, reverse=False)[0]
        rec_e = net(lr=lr_e, reverse=True, eps_std=0.0)
        net.set_precision("f16x3")
        try:
            eng = net.engine()
            eng.profile_convs(True)
            lr_f = net(hr=hr, reverse=False)[0]
            _, n7, _, _ = eng.conv_time(9, 1, kind=7, reset=True)
            rec_f = net(lr=lr_e, reverse=True, eps_std=0.0)
            _, m7, _, _ = eng.conv_time(9, 1, kind=7, reset=True)
            eng.profile_convs(False)
            assert eng.fallback_count() == 0
        finally:
            net.set_precision("exact")
    assert (n7, m7) == (2 * n_rdb, 2 * n_rdb), (n7, m7, n_rdb)
    assert maxdiff(lr_f, lr_e) <= 1.0 / 255 + 1e-6                       # quantised LR^: at most a single level flips
    assert float(((lr_f - lr_e).abs() > 1e-6).float().mean()) < 0.01
    assert maxdiff(rec_f, rec_e) <= 2e-5 * max(1.0, float(rec_e.abs().max()))


def test_inference_after_an_optimiser_step_keeps_the_fat_schedule():
    """train_HCFlow.py:208-305 validates between optimiser steps. After `opt.step()` the engine refreshes its packs on the device
    (hcf_refresh_from_device); round 6 rebuilds the DERIVED Winograd packs there too -- fat dense-block pairs, the conditional
    FCN's padded conv1 + conv2 fragment pack -- so the following eval-mode inverse call (i) still takes the fat launches (conv
    profile kinds 7 and 6) and (ii) equals a FRESH module fed the updated parameters through the host path (hcf_finalize)."""
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import eps_shapes
    from tests.test_gpu_backward import _fresh_sr
    from tests.util import maxdiff
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(19)
    B, size = 2, 40
    hr = torch.rand(B, 3, 4 * size, 4 * size, generator=g).cuda()
    lr = torch.rand(B, 3, size, size, generator=g).cuda()
    noise = torch.rand(hr.shape, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.8 for s in eps_shapes(cfg, B, size, size)]
    opt = torch.optim.SGD(net.parameters(), lr=1e-7)
    for _ in range(3):                                  # step 1: host path; steps 2, 3 see a device refresh
        opt.zero_grad()
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        opt.step()
    net.eval()
    net.set_precision("f16x3")
    try:
        with torch.no_grad():
            a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)      # refresh happens here
            eng = net.engine()
            eng.profile_convs(True)
            b = net.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            n_fat = eng.conv_time(9, 0, kind=7)[1]
            n_fcn = eng.conv_time(9, 0, kind=6, reset=True)[1]
            eng.profile_convs(False)
        assert net._engines[0]["ptrs"] is not None       # (the device-side path, not a host re-finalise)
        assert n_fat > 0 and n_fcn > 0, (n_fat, n_fcn)
        assert torch.equal(a, b)
        ref = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
        for m in ref.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        ref = ref.to("cuda:0").eval().set_precision("f16x3")
        with torch.no_grad():
            c = ref.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            e2 = ref.engine()
            e2.profile_convs(True)
            ref.reverse_flow_diracLR(lr, None, None, eps_std=0.8, eps=eps, clamp=False)
            assert e2.conv_time(9, 0, kind=7, reset=True)[1] == n_fat       # the same schedule on both sides
            e2.profile_convs(False)
        # (the device refresh evaluates exp(logs) with the GPU's expf, the host path with libm: ~1 ulp in the epilogue scales)
        assert maxdiff(a, c) <= 2e-5 * max(1.0, float(c.abs().max()))
    finally:
        net.set_precision("exact")


def test_in_place_updates_of_a_rescaling_net_keep_its_winograd_schedules():
    """The rescaling nets' derived packs -- 16-channel fat pairs (one 32-channel launch + a zero-padded completion) and the
    DenseBlock convs over [z1 padded | growth] -- are rebuilt on the device as well: after an in-place update of every parameter
    the round trip equals a FRESH module's and the launch mix (Winograd kinds 4 / 7) is the same as before the update."""
    from hcflow_amd import HCFlowNet_Rescaling
    from hcflow_amd.config import preset
    from tests.util import cached_params, maxdiff
    cfg = preset("Rescaling_4X_tiny")
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params("Rescaling_4X_tiny", 13), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").eval().set_precision("f16x3")
    g = torch.Generator().manual_seed(23)
    hr = torch.rand(2, 3, 96, 128, generator=g).cuda()

    def roundtrip(n):
        lr_hat, _, _ = n(hr=hr, reverse=False)
        lrq = (torch.clamp(lr_hat, 0, 1) * 255.).round() / 255.
        return lr_hat, n(lr=lrq, eps_std=0.0, reverse=True)

    def mix(n):
        e = n.engine()
        e.profile_convs(True)
        roundtrip(n)
        r = (e.conv_time(9, 0, kind=4)[1], e.conv_time(9, 0, kind=7, reset=True)[1])
        e.profile_convs(False)
        return r
    with torch.no_grad():
        roundtrip(net)                                   # host path, binds pointers
        before = mix(net)
        for p in net.parameters():
            if p.requires_grad:
                p.mul_(1.0 + 1e-3)                       # in place: only _version moves -> device-side refresh
        a = roundtrip(net)
        after = mix(net)
    assert net._engines[0]["ptrs"] is not None
    assert before[0] > 0 and before == after, (before, after)
    ref = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    for m in ref.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    ref = ref.to("cuda:0").eval().set_precision("f16x3")
    with torch.no_grad():
        b = roundtrip(ref)
        assert mix(ref) == after
    for x, y in zip(a, b):
        assert maxdiff(x, y) <= 2e-5 * max(1.0, float(y.abs().max()))
