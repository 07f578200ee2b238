"""-m gpu: backward kernels of the training path against torch.autograd on the CPU
(fp64 evaluation of the same op)."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_gpu_ops import _gen

pytestmark = pytest.mark.gpu

# B, H, W, source channels, source upsample log2, cout, k
BWD_CASES = [
    (2, 16, 32, [32], [0], 32, 3),
    (1, 24, 40, [64, 96], [0, 0], 32, 3),          # RDB conv4 shape, ragged tile edges
    (2, 16, 16, [64, 128], [0, 0], 64, 3),         # RDB conv5
    (1, 16, 32, [6, 64], [0, 0], 64, 3),           # FCN conv1: 6 + cond channels
    (1, 16, 32, [3, 128], [0, 1], 64, 3),          # conv_first: second source read through an upsample
    (2, 12, 20, [64], [0], 64, 1),                 # FCN conv2 (1x1)
    (1, 9, 33, [64], [0], 12, 3),                  # Conv2dZeros head, 12 outputs, odd sizes
    (1, 16, 32, [21, 10], [0, 0], 22, 3),          # channel tails
]


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(1e-30, float(b.double().abs().max())))


@pytest.mark.parametrize("case", BWD_CASES)
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_conv2d_backward(case, precision):
    from hcflow_amd import ops
    B, H, W, cs, ups, cout, k = case
    g = _gen(sum(cs) + cout + H)
    srcs = [torch.randn(B, c, H >> u, W >> u, generator=g) for c, u in zip(cs, ups)]
    cin = sum(cs)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    gy = torch.randn(B, cout, H, W, generator=g)
    # reference: fp64 autograd
    s64 = [s.double().requires_grad_(True) for s in srcs]
    w64 = w.double().requires_grad_(True)
    b64 = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    x = torch.cat([F.interpolate(s, scale_factor=2 ** u, mode="nearest") if u else s for s, u in zip(s64, ups)], 1)
    y = F.conv2d(x, w64, b64, 1, k // 2)
    y.backward(gy.double())
    ops.set_precision(precision)
    try:
        dsrcs, dw, db = ops.conv2d_backward([s.cuda() for s in srcs], w, gy.cuda(), ups)
    finally:
        ops.set_precision("exact")
    for d, s in zip(dsrcs, s64):
        assert _rel(d.cpu(), s.grad) <= 3e-6, (case, precision, _rel(d.cpu(), s.grad))
    assert _rel(dw, w64.grad) <= 3e-6, (case, _rel(dw, w64.grad))
    assert _rel(db, b64.grad) <= 1e-6


@pytest.mark.parametrize("shape", [(16, 40, 40, [64, 32], [0, 0], 32), (5, 20, 20, [32], [0], 64), (3, 9, 33, [64], [0], 32),
                                   (4, 16, 24, [16, 32], [0, 1], 32), (7, 8, 1, [32], [0], 32)])
def test_conv2d_weight_gradient_f16x3_strips(shape, monkeypatch):
    """Narrow images: the f16x3 weight-gradient kernel walks STRIPS (the B images of a tile row side by side with one zero
    column between them, WgradArgs::strip_w) instead of every image's own tiles -- the same sum over pixels in another order.
    Against fp64, against the per-image walk (HCF_NO_WG_STRIP=1), and reproducible."""
    from hcflow_amd import ops
    B, H, W, cs, ups, cout = shape
    g = _gen(B * 1000 + W)
    srcs = [torch.randn(B, c, H >> u, W >> u, generator=g) for c, u in zip(cs, ups)]
    gy = torch.randn(B, cout, H, W, generator=g)
    s64 = [t.double().requires_grad_(True) for t in srcs]
    x = torch.cat([F.interpolate(t, scale_factor=2 ** u, mode="nearest") if u else t for t, u in zip(s64, ups)], 1)
    ref = torch.nn.grad.conv2d_weight(x.detach(), (cout, sum(cs), 3, 3), gy.double(), stride=1, padding=1)
    w = torch.randn(cout, sum(cs), 3, 3, generator=g) / (9 * sum(cs)) ** 0.5
    F.conv2d(x, w.double(), None, 1, 1).backward(gy.double())
    ops.set_precision("f16x3")
    try:
        res = []
        for off in (False, False, True):
            for k in ("HCF_NO_WG_STRIP", "HCF_NO_DG_STRIP", "HCF_NO_DG_TH4"):      # the scaled data-gradient conv walks strips as well,
                # and takes 4-row tiles while the grid stays within one block per CU
                if off:
                    monkeypatch.setenv(k, "1")
                else:
                    monkeypatch.delenv(k, raising=False)
            ds, dw, _ = ops.conv2d_backward([t.cuda() for t in srcs], w, gy.cuda(), ups)
            res.append((dw.cpu(), [d.cpu() for d in ds]))
    finally:
        ops.set_precision("exact")
    assert torch.equal(res[0][0], res[1][0])
    assert _rel(res[0][0], ref) <= 3e-6 and _rel(res[2][0], ref) <= 3e-6
    assert _rel(res[0][0], res[2][0]) <= 2e-6
    for d0, d1, d2, t in zip(res[0][1], res[1][1], res[2][1], s64):
        assert torch.equal(d0, d1)
        assert torch.equal(d0, d2)                 # per output pixel the same products in the same order: bit-equal to the per-image walk
        assert _rel(d0, t.grad) <= 3e-6


@pytest.mark.parametrize("gscale", [1e-9, 1.0, 3e4])
def test_conv2d_weight_gradient_f16x3_ranges(gscale):
    """The f16 matrix-core weight gradient (hcf_conv_wgrad.hip, conv_wgrad_f16x3_kernel) scales G by a power of two taken
    from max |g|: gradients of 1e-9 and of 3e4 keep fp32-class accuracy, and so do activations spanning six decades
    (several tiles per block, ragged edges, two sources)."""
    from hcflow_amd import ops
    g = _gen(77)
    B, H, W, cs, cout = 3, 40, 72, [64, 32], 32
    srcs = [torch.randn(B, c, H, W, generator=g) * torch.logspace(-3, 3, c).view(1, c, 1, 1) for c in cs]
    gy = torch.randn(B, cout, H, W, generator=g) * gscale
    gy[:, :, : H // 2] *= 1e-3                                   # a quiet half: 1000x below the scaling reference
    x = torch.cat(srcs, 1).double()
    ref = torch.nn.grad.conv2d_weight(x, (cout, sum(cs), 3, 3), gy.double(), stride=1, padding=1)
    ops.set_precision("f16x3")
    try:
        _, dw, _ = ops.conv2d_backward([s.cuda() for s in srcs], torch.zeros(cout, sum(cs), 3, 3), gy.cuda(), need_input_grads=False)
        _, dw2, _ = ops.conv2d_backward([s.cuda() for s in srcs], torch.zeros(cout, sum(cs), 3, 3), gy.cuda(), need_input_grads=False)
    finally:
        ops.set_precision("exact")
    assert torch.equal(dw, dw2)                                   # fixed reduction order
    # per input channel (their magnitudes differ by 1e6): error relative to that channel's largest entry
    err = (dw.double() - ref).abs().amax(dim=(0, 2, 3)) / ref.abs().amax(dim=(0, 2, 3))
    assert float(err.max()) <= 5e-6, float(err.max())


@pytest.mark.parametrize("shape", [(16, 40, 40, [64, 32], 32, 3), (3, 40, 72, [64, 32], 64, 3), (2, 80, 96, [32], 32, 3),
                                   (4, 64, 64, [64, 64, 64], 64, 3), (8, 64, 96, [64, 64, 64], 64, 3), (1, 8, 32, [32], 32, 3),
                                   (5, 20, 20, [21], 12, 3), (4, 48, 64, [64], 64, 1), (6, 64, 96, [64, 64, 64], 64, 1),
                                   (3, 9, 33, [48], 24, 1)])
def test_conv2d_weight_gradient_f16x3_lds_forms_are_bit_identical(shape, monkeypatch):
    """The f16x3 weight-gradient kernels stage their tiles through two LDS buffers, one staging slot converted, written and
    reloaded per tap in the shadow of that tap's MFMAs (the default); HCF_WG_DB_BLOCK=1 writes a tile's slots in one block,
    HCF_WG_SINGLE_BUF=1 is the single-buffer form (barrier, convert + write, barrier, MFMAs). All three run the same MFMAs
    on the same fragments in the same order: the results are bit-identical (one, two, four and ten tiles per block, ragged
    tiles, channel tails, strips, 1x1), and right against fp64."""
    from hcflow_amd import ops
    B, H, W, cs, cout, k = shape
    g = _gen(31 * B + W + k)
    srcs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    gy = torch.randn(B, cout, H, W, generator=g)
    ref = torch.nn.grad.conv2d_weight(torch.cat(srcs, 1).double(), (cout, sum(cs), k, k), gy.double(), stride=1, padding=k // 2)
    w0 = torch.zeros(cout, sum(cs), k, k)
    ops.set_precision("f16x3")
    try:
        res = {}
        for form in ("interleaved", "block", "single"):
            monkeypatch.delenv("HCF_WG_DB_BLOCK", raising=False)
            monkeypatch.delenv("HCF_WG_SINGLE_BUF", raising=False)
            if form == "block":
                monkeypatch.setenv("HCF_WG_DB_BLOCK", "1")
            if form == "single":
                monkeypatch.setenv("HCF_WG_SINGLE_BUF", "1")
            for _ in range(2):
                _, dw, _ = ops.conv2d_backward([t.cuda() for t in srcs], w0, gy.cuda(), need_input_grads=False)
                res.setdefault(form, []).append(dw.cpu())
    finally:
        ops.set_precision("exact")
    for form in res:
        assert torch.equal(res[form][0], res[form][1]), form       # reproducible (no race between the two LDS buffers)
    assert torch.equal(res["interleaved"][0], res["single"][0])
    assert torch.equal(res["block"][0], res["single"][0])
    assert _rel(res["interleaved"][0], ref) <= 3e-6


GRADS = ["grad_sr4_tiny", "grad_sr8_tiny"]


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", GRADS)
def test_nll_step_gradients_match_reference(name, precision):
    """One NLL step of HCFlow_SR_model.optimize_parameters (:195-199) through the drop-in module: nll and
    d nll / d parameter for every tensor of the net against the reference-generated fixture."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import param_spec
    from tests.util import load_golden, params_for, t
    from tests.test_oracle_golden import check_grads_against_fixture
    g = load_golden(name)
    cfg, p = params_for(g)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train().set_precision(precision)     # f16x3: forward + data-gradient 3x3 convs on the split kernels
    lr_hat, nll = net(hr=t(g["hr"]).cuda(), lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
    assert abs(float(nll.detach()) - float(g["fwd_nll"])) <= 2e-4 * max(1.0, abs(float(g["fwd_nll"])) / 100)
    assert float((lr_hat.cpu() - t(g["fwd_lr"])).abs().max()) <= 1e-4
    (nll * 1.0).backward()
    sd = dict(net.named_parameters())
    grads = [np.zeros(tuple(sd[k].shape), np.float32) if sd[k].grad is None else sd[k].grad.cpu().numpy()
             for k, _, _ in param_spec(cfg)]
    assert all(np.isfinite(x).all() for x in grads)
    assert net.engine().fallback_count() == 0
    check_grads_against_fixture(g, grads)


def test_nll_step_gradients_are_bit_identical_across_the_weight_gradient_lds_forms(monkeypatch):
    """The engine's batched weight-gradient launch (the five convs of a dense block in one grid) and its one-conv launches under
    the three LDS forms of the f16x3 kernel: every parameter gradient of one NLL step is bit-identical."""
    from tests.util import load_golden, params_for, t
    from hcflow_amd import HCFlowNet_SR
    g = load_golden("grad_sr4_tiny")
    cfg, p = params_for(g)
    got = {}
    for form in ("interleaved", "block", "single"):
        monkeypatch.delenv("HCF_WG_DB_BLOCK", raising=False)
        monkeypatch.delenv("HCF_WG_SINGLE_BUF", raising=False)
        if form == "block":
            monkeypatch.setenv("HCF_WG_DB_BLOCK", "1")
        if form == "single":
            monkeypatch.setenv("HCF_WG_SINGLE_BUF", "1")
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(p, strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train().set_precision("f16x3")
        _, nll = net(hr=t(g["hr"]).cuda(), lr=t(g["lr"]).cuda(), reverse=False, noise=t(g["fwd_noise"]).cuda())
        nll.backward()
        torch.cuda.synchronize()
        got[form] = {k: v.grad.cpu().clone() for k, v in net.named_parameters() if v.grad is not None}
    assert len(got["single"]) > 50
    for k, v in got["single"].items():
        assert torch.equal(got["interleaved"][k], v), k
        assert torch.equal(got["block"][k], v), k


def _fresh_sr(name, seed, inited=True):
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params
    cfg = preset(name)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params(name, seed), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = inited
    return cfg, net.to("cuda:0")


def test_training_steps_with_adam_reduce_nll_and_track_parameter_updates():
    """A few optimiser steps as HCFlow_SR_model.optimize_parameters runs them (:184-205: nll.backward(), gradient
    clipping, Adam): the loss falls on a fixed batch, every parameter gets a finite gradient, and the engine sees
    the in-place parameter updates (the second forward differs from the first)."""
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(5)
    hr = torch.rand(2, 3, 64, 64, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4, betas=(0.9, 0.99))
    losses = []
    for it in range(4):
        opt.zero_grad()
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.requires_grad)
        torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)
        opt.step()
        losses.append(float(nll.detach()))
    assert losses[-1] < losses[0], losses
    with torch.no_grad():
        net.eval()
        _, nll_eval = net(hr=hr, lr=lr, reverse=False, noise=noise)
    assert float(nll_eval) < losses[0]


def test_first_training_step_fits_actnorms_then_differentiates():
    """train() mode, un-initialised ActNorms, autograd on: the ActNorms are fitted from this batch first (as the
    reference does inside the same forward, ActNorms.py:78-80) and the returned nll / gradients are those of the
    fitted net (checked against the oracle's autograd on the same inputs)."""
    import numpy as np
    from oracle import hcflow_oracle as O
    from hcflow_amd.config import param_spec
    from tests.util import cached_params
    cfg, net = _fresh_sr("SR_4X_tiny", 11, inited=False)
    an = [(k, m) for k, m in net.named_modules() if "ActNorm" in type(m).__name__]
    with torch.no_grad():
        for _, m in an:
            m.bias.zero_()
            m.logs.zero_()
    net.train()
    g = torch.Generator().manual_seed(6)
    hr = torch.rand(2, 3, 48, 64, generator=g)
    lr = torch.rand(2, 3, 12, 16, generator=g)
    noise = torch.rand(hr.shape, generator=g)
    _, nll = net(hr=hr.cuda(), lr=lr.cuda(), reverse=False, noise=noise.cuda())
    nll.backward()
    assert all(m.inited for _, m in an)
    # oracle: fit on the same batch, then differentiate with the fitted values held fixed for the fit itself
    p0 = {k: v.clone() for k, v in cached_params("SR_4X_tiny", 11).items()}
    for k, _ in an:
        p0[k + ".bias"] = torch.zeros_like(p0[k + ".bias"])
        p0[k + ".logs"] = torch.zeros_like(p0[k + ".logs"])
    ip = O.InitParams(p0, [k for k, _ in an])
    with torch.no_grad():
        O.sr_forward(hr, lr, ip, cfg, noise=noise)
    q = {k: v.clone().requires_grad_(True) for k, v in ip.items()}
    _, nll_o = O.sr_forward(hr, lr, q, cfg, noise=noise)
    nll_o.backward()
    assert abs(float(nll.detach()) - float(nll_o.detach())) <= 2e-4 * max(1.0, abs(float(nll_o.detach())) / 100)
    sd = dict(net.named_parameters())
    gmax = max(float(q[k].grad.norm()) for k, _, _ in param_spec(cfg) if q[k].grad is not None)
    for k, _, _ in param_spec(cfg):
        ref = q[k].grad
        if ref is None:
            continue
        err = float((sd[k].grad.cpu() - ref).norm())
        assert err <= 3e-4 * max(float(ref.norm()), 1e-6 * gmax), (k, err, float(ref.norm()))


def test_ddp_wrapped_module_trains():
    """nn.parallel.DistributedDataParallel(netG, device_ids=[dev]) as HCFlow_SR_model does (:33-36), world size 1
    over RCCL: forward by keyword, backward, gradients identical to the unwrapped module."""
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(7)
    hr = torch.rand(2, 3, 32, 32, generator=g).cuda()
    lr = torch.rand(2, 3, 8, 8, generator=g).cuda()
    noise = torch.rand(hr.shape, generator=g).cuda()
    _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
    nll.backward()
    want = [p.grad.clone() for p in net.parameters() if p.requires_grad]
    net.zero_grad()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ddp = DDP(net, device_ids=[0])
        _, nll2 = ddp(hr=hr, lr=lr, reverse=False, noise=noise)
        nll2.backward()
        got = [p.grad for p in net.parameters() if p.requires_grad]
        assert abs(float(nll2.detach()) - float(nll.detach())) <= 1e-6 * max(1.0, abs(float(nll.detach())))
        # relative to each tensor's largest entry, with the noise floor of check_grads_against_fixture: tensors 1e-5 of the net's
        # largest gradient are sums of much larger cancelling terms (f16x3: the split's absolute floor, 1e-9 here)
        gmax = max(float(b.abs().max()) for b in want)
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-4 * max(2e-5 * gmax, float(b.abs().max()))
        # the reference builds its optimiser AFTER the DDP wrap (HCFlow_SR_model.py:33-36, then :118): the one-launch Adam re-points
        # the parameters into its flat buffer underneath DDP; two steps, the loss moves, gradients keep arriving in DDP's buckets
        from hcflow_amd import optim as hopt
        opt = hopt.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-5, betas=(0.9, 0.99))
        losses = []
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            _, l = ddp(hr=hr, lr=lr, reverse=False, noise=noise)
            l.backward()
            hopt.clip_grad_norm_(net.parameters(), 100.0)
            opt.step()
            losses.append(float(l.detach()))
        assert abs(losses[0] - float(nll.detach())) <= 1e-6 * max(1.0, abs(losses[0]))
        assert len(set(losses)) == 3 and all(l_ == l_ for l_ in losses)
        assert len({p.untyped_storage().data_ptr() for p in net.parameters() if p.requires_grad}) == 1
    finally:
        dist.destroy_process_group()


def test_device_refresh_equals_host_repack():
    """After an in-place parameter update the engine refreshes its packs on the device (hcf_refresh_from_device);
    forward, inverse (both precisions) and the next gradients must equal those of a freshly built engine fed the same
    parameters through the host path (hcf_set_param + hcf_finalize)."""
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import eps_shapes
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(9)
    hr = torch.rand(2, 3, 64, 96, generator=g).cuda()
    lr = torch.rand(2, 3, 16, 24, generator=g).cuda()
    noise = torch.rand(hr.shape, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.7 for s in eps_shapes(cfg, 2, 16, 24)]
    opt = torch.optim.SGD(net.parameters(), lr=1e-7)       # gradients reach 1e4 on these random weights
    for _ in range(2):                                  # step 1: host path; step 2 sees a device refresh
        opt.zero_grad()
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        opt.step()
    # third pass on the refreshed engine
    opt.zero_grad()
    _, nll_a = net(hr=hr, lr=lr, reverse=False, noise=noise)
    nll_a.backward()
    grads_a = [p.grad.clone() for p in net.parameters() if p.requires_grad]
    with torch.no_grad():
        inv_a = net.reverse_flow_diracLR(lr, None, None, eps_std=0.7, eps=eps, clamp=False)
        net.set_precision("f16x3")
        inv_a16 = net.reverse_flow_diracLR(lr, None, None, eps_std=0.7, eps=eps, clamp=False)
        net.set_precision("exact")
    assert net._engines[0]["ptrs"] is not None
    # reference: a new module / engine with the same parameter values through the host path
    ref = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=True)
    for m in ref.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    ref = ref.to("cuda:0").train()
    _, nll_b = ref(hr=hr, lr=lr, reverse=False, noise=noise)
    nll_b.backward()
    grads_b = [p.grad for p in ref.parameters() if p.requires_grad]
    assert float(nll_a.detach()) == float(nll_b.detach())
    assert all(bool(torch.isfinite(x).all()) for x in grads_a)
    gmax = max(float(b.abs().max()) for b in grads_b)
    for a, b in zip(grads_a, grads_b):                  # 1-ulp expf differences in the scales + atomically summed bias grads
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-8 * gmax
    with torch.no_grad():
        inv_b = ref.reverse_flow_diracLR(lr, None, None, eps_std=0.7, eps=eps, clamp=False)
        ref.set_precision("f16x3")
        inv_b16 = ref.reverse_flow_diracLR(lr, None, None, eps_std=0.7, eps=eps, clamp=False)
    # (the device refresh evaluates exp(logs) with the GPU's expf, the host path with libm: ~1 ulp in the epilogue scales)
    scale = max(1.0, float(inv_b.abs().max()))
    assert float((inv_a - inv_b).abs().max()) <= 2e-5 * scale
    assert float((inv_a16 - inv_b16).abs().max()) <= 2e-5 * scale


RGRADS = ["rgrad_sr4_tiny", "rgrad_sr8_tiny"]


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", RGRADS)
def test_reverse_path_gradients_match_reference(name, precision):
    """fake_H = netG(lr=, eps_std=, reverse=True); L1(fake_H, real_H).backward() (HCFlow_SR_model.py:207-216, the HR
    pixel loss of the HCFlow+ / ++ recipes) through the drop-in module: fake_H, the loss and d loss / d parameter for
    every tensor against the reference-generated fixture (same captured eps)."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import param_spec
    from tests.util import load_golden, params_for, t
    from tests.test_oracle_golden import check_grads_against_fixture, rgrad_eps
    g = load_golden(name)
    cfg, p = params_for(g)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train().set_precision(precision)
    fake = net(lr=t(g["lr"]).cuda(), z=None, u=None, eps_std=float(g["tau"]), reverse=True,
               eps=[e.cuda() for e in rgrad_eps(g)])
    assert float((fake.detach().cpu() - t(g["fake"])).abs().max()) <= 1e-4
    loss = F.l1_loss(fake, t(g["hr"]).cuda())
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-5
    loss.backward()
    sd = dict(net.named_parameters())
    grads = [np.zeros(tuple(sd[k].shape), np.float32) if sd[k].grad is None else sd[k].grad.cpu().numpy()
             for k, _, _ in param_spec(cfg)]
    assert all(np.isfinite(x).all() for x in grads)
    check_grads_against_fixture(g, grads, rtol=5e-4)


def test_hcflow_plus_style_step_mixes_both_tapes_and_a_torch_discriminator():
    """The HCFlow+ / ++ generator step (HCFlow_SR_model.optimize_parameters :189-255): NLL backward + optimiser step,
    then the reverse pass at eps_std = 0 with an L1 pixel loss, then a sampled fake_H scored by a stock-PyTorch
    discriminator; every phase gives finite gradients and the pixel phase lowers its loss."""
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train().set_precision("f16x3")
    g = torch.Generator().manual_seed(15)
    hr = torch.rand(2, 3, 64, 64, generator=g).cuda() * 0.6 + 0.2
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    disc = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, 2, 1), torch.nn.LeakyReLU(0.2), torch.nn.Conv2d(8, 1, 3, 2, 1)).cuda()
    opt = torch.optim.Adam(net.parameters(), lr=1e-5)
    pix = []
    try:
        for it in range(3):
            opt.zero_grad()
            _, nll = net(hr=hr, lr=lr, u=None, reverse=False)
            (1e-3 * nll).backward()
            opt.step()
            opt.zero_grad()
            fake = net(lr=lr, z=None, u=None, eps_std=0.0, reverse=True)
            l_pix = F.l1_loss(fake, hr)
            l_pix.backward()
            assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
            opt.step()
            pix.append(float(l_pix.detach()))
            opt.zero_grad()
            fake = net(lr=lr, z=None, u=None, eps_std=0.8, reverse=True)
            l_gan = F.softplus(-disc(fake)).mean()                 # generator side of a non-saturating GAN loss
            l_gan.backward()
            assert any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in net.parameters())
            assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
        assert pix[-1] < pix[0], pix
    finally:
        net.set_precision("exact")


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_rescaling_step_gradients_match_reference(precision):
    """One generator step of HCFlow_Rescaling_model.optimize_parameters (:212-256): forward -> losses on fake_LR / z,
    Quant (straight-through), inverse -> L1 on fake_H, ONE backward through both taped passes (two tape slots,
    gradient w.r.t. the inverse pass's LR input) against the reference-generated fixture."""
    import numpy as np
    from hcflow_amd import HCFlowNet_Rescaling
    from hcflow_amd.config import param_spec
    from tests.util import load_golden, params_for, t
    from tests.test_oracle_golden import check_grads_against_fixture, rgrad_eps, rescale_step_loss
    g = load_golden("grad_rescale_tiny")
    cfg, p = params_for(g)
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    net.load_state_dict(p, strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train().set_precision(precision)
    eps = [e.cuda() for e in rgrad_eps(g)]
    l_lr, l_z, l_hr, fake_lr, fake_h = rescale_step_loss(
        lambda x: net(hr=x, u=None, reverse=False),
        lambda x, e: net(lr=x, z=None, u=None, eps_std=1.0, reverse=True, eps=e),
        t(g["hr"]).cuda(), t(g["lr"]).cuda(), eps)
    assert float((fake_lr.detach().cpu() - t(g["fake_lr"])).abs().max()) <= 1e-4
    assert float((fake_h.detach().cpu() - t(g["fake_h"])).abs().max()) <= 1e-4
    assert abs(float(l_hr.detach()) - float(g["l_hr"])) <= 1e-5 and abs(float(l_lr.detach()) - float(g["l_lr"])) <= 1e-7
    (l_lr + l_z + l_hr).backward()
    sd = dict(net.named_parameters())
    grads = [np.zeros(tuple(sd[k].shape), np.float32) if sd[k].grad is None else sd[k].grad.cpu().numpy()
             for k, _, _ in param_spec(cfg)]
    assert all(np.isfinite(x).all() for x in grads)
    # 5e-3: in this fixture ONE pre-activation of level0_condFlow.additional_flow_steps.1.affine.f.conv1 lies within fp32
    # summation noise of zero, so its ReLU mask differs between implementations (one pixel's term in one output
    # channel's bias / weight-row gradient, 0.3 % of those tensors); everything else agrees to 1e-6
    # (tools/dbg_rescale_grads.py compares against the oracle's autograd tensor by tensor).
    check_grads_against_fixture(g, grads, rtol=5e-3, elem_rtol=3e-2)


def test_rescaling_training_loop_runs_and_improves():
    """A few generator steps of HCFlow_Rescaling_model.optimize_parameters (:204-256) with Adam on a fixed batch, first
    step with un-initialised ActNorms (train() mode): finite gradients everywhere, the HR reconstruction loss falls."""
    from hcflow_amd import HCFlowNet_Rescaling
    from hcflow_amd.config import preset
    from tests.util import cached_params
    from tests.test_oracle_golden import rescale_step_loss
    cfg = preset("Rescaling_4X_tiny")
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params("Rescaling_4X_tiny", 13), strict=True)
    an = [m for m in net.modules() if "ActNorm" in type(m).__name__]
    with torch.no_grad():
        for m in an:
            m.bias.zero_()
            m.logs.zero_()
    net = net.to("cuda:0").train().set_precision("f16x3")
    g = torch.Generator().manual_seed(31)
    hr = torch.rand(2, 3, 64, 96, generator=g).cuda() * 0.8 + 0.1
    lr = F.avg_pool2d(hr, 4)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4)
    hist = []
    try:
        for it in range(4):
            opt.zero_grad()
            l_lr, l_z, l_hr, _, _ = rescale_step_loss(
                lambda x: net(hr=x, u=None, reverse=False),
                lambda x, e: net(lr=x, z=None, u=None, eps_std=1.0, reverse=True), hr, lr, None)
            (l_lr + l_z + l_hr).backward()
            assert all(m.inited for m in an)
            assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)
            torch.nn.utils.clip_grad_norm_(net.parameters(), 10.0)
            opt.step()
            hist.append(float(l_hr.detach()))
        assert hist[-1] < hist[0], hist
    finally:
        net.set_precision("exact")


def test_training_api_error_paths():
    """Misuse is reported, never silently computed: a second backward on the same graph, odd sizes (squeeze2d's
    assert, Basic.py:136), autograd through a path that has no backward."""
    from hcflow_amd import _lib
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(2)
    hr = torch.rand(1, 3, 32, 32, generator=g).cuda()
    lr = torch.rand(1, 3, 8, 8, generator=g).cuda()
    _, nll = net(hr=hr, lr=lr, reverse=False)
    nll.backward(retain_graph=True)
    with pytest.raises(_lib.HcfError):                 # the engine's tape is consumed by the first backward
        nll.backward()
    with pytest.raises((_lib.HcfError, AssertionError)):
        net(hr=torch.rand(1, 3, 30, 32).cuda(), lr=lr, reverse=False)
    # a taped forward whose backward never runs must not poison later passes
    _, nll2 = net(hr=hr, lr=lr, reverse=False)
    with torch.no_grad():
        out = net(lr=lr, eps_std=0.5, reverse=True)
    assert bool(torch.isfinite(out).all())
    nll2.backward()                                    # inference in between used its own arena: the tape is intact
    assert all(bool(torch.isfinite(p.grad).all()) for p in net.parameters() if p.grad is not None)


def test_weight_gradient_f16x3_falls_back_when_the_input_leaves_the_f16_range():
    """the f16x3 weight-gradient kernel splits X without a scale, so an |x| >= 65504 that no forward range check
    covered (1x1 convs, > 64 output channels, the per-op entry point) must take the fp32 kernel: finite and equal to exact."""
    from hcflow_amd import ops
    g = _gen(5)
    x = torch.randn(2, 32, 24, 40, generator=g)
    x[1, 7, 3, 9] = 7.0e4
    gy = torch.randn(2, 32, 24, 40, generator=g)
    w0 = torch.zeros(32, 32, 3, 3)
    _, dw_exact, _ = ops.conv2d_backward([x.cuda()], w0, gy.cuda(), need_input_grads=False)
    ops.set_precision("f16x3")
    try:
        _, dw, _ = ops.conv2d_backward([x.cuda()], w0, gy.cuda(), need_input_grads=False)
    finally:
        ops.set_precision("exact")
    assert bool(torch.isfinite(dw).all()) and torch.equal(dw, dw_exact)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_dense_block_gather_form_data_gradients_equal_the_per_conv_form(precision, monkeypatch):
    """The dense blocks' data gradients run in GATHER form (one conv per tensor x_m over the gradients of every later conv of the
    block, hcf_engine_train.inc make_rdb_gather_packs) instead of one short-K conv per (conv, source window): the same sums in
    another order. Both forms on the same step (HCF_NO_DGRAD_GATHER=1 selects the per-conv form when an engine prepares for
    training), before and after an optimiser step (device-side refresh of the composite packs)."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(21)
    hr = torch.rand(2, 3, 96, 128, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("HCF_NO_DGRAD_GATHER", "1")
        else:
            monkeypatch.delenv("HCF_NO_DGRAD_GATHER", raising=False)
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train().set_precision(precision)
        opt = torch.optim.SGD([q for q in net.parameters() if q.requires_grad], lr=1e-7)      # gradients reach 1e4 on these random weights
        steps = []
        for it in range(2):
            opt.zero_grad(set_to_none=True)
            _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
            nll.backward()
            steps.append((float(nll.detach()), spec_grads(net, cfg)))
            opt.step()
        res.append(steps)
    for it in range(2):
        (n0, g0), (n1, g1) = res[0][it], res[1][it]
        assert abs(n0 - n1) <= 1e-6 * abs(n1)
        gmax = max(float(np.abs(x).max()) for x in g1)
        for a, b in zip(g0, g1):
            assert np.isfinite(a).all()
            assert float(np.abs(a - b).max()) <= 2e-4 * max(float(np.abs(b).max()), 2e-5 * gmax)


def test_fused_epilogue_backward_in_the_gather_convs_equals_the_separate_kernel(monkeypatch):
    """f16x3 training: the gather conv of x_m (m >= 1) applies the epilogue backward of the conv that produced x_m in its own
    epilogue (ConvArgs::fb_y, hcf_engine_train.inc rdb_fuse_ok). dL/dpre is the same fp32 product either way, so weight gradients
    are equal bit for bit; bias / ActNorm gradients are sums over differently shaped blocks (rounding only). HCF_NO_EPI_FUSE=1 selects
    the separate conv_epilogue_bwd launch (HCF_NO_FCN_FUSE=1: in the FCN chains only), read at the start of each backward pass."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(33)
    hr = torch.rand(2, 3, 96, 160, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    net = net.to("cuda:0").train().set_precision("f16x3")
    res = []
    for off in (False, True, False):
        for k in ("HCF_NO_EPI_FUSE", "HCF_NO_FCN_FUSE"):      # (the FCNs' conv2 / conv3 data gradients apply conv1's / conv2's likewise)
            if off:
                monkeypatch.setenv(k, "1")
            else:
                monkeypatch.delenv(k, raising=False)
        net.zero_grad(set_to_none=True)
        _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        nll.backward()
        res.append((float(nll.detach()), spec_grads(net, cfg)))
    (n0, g0), (n1, g1), (n2, g2) = res
    assert n0 == n1 == n2
    for a, c in zip(g0, g2):
        assert np.array_equal(a, c)                       # the fused form is reproducible run to run
    ndiff = 0
    for a, b in zip(g0, g1):
        assert np.isfinite(a).all()
        if a.size != max(a.shape):                        # conv / invconv weights: products of the same dL/dpre
            assert np.array_equal(a, b)
        else:                                             # vectors (conv biases, ActNorm bias / logs as [1, C, 1, 1]): sums over other blocks
            ndiff += int(not np.array_equal(a, b))
            assert float(np.abs(a - b).max()) <= 2e-6 * max(float(np.abs(b).max()), 1e-30)
    assert ndiff > 0                                      # the knob did select another summation order


@pytest.mark.parametrize("fused", [True, False])
def test_winograd_gather_data_gradients_equal_the_direct_form(fused, monkeypatch):
    """f16x3 training: above HCF_DGRAD_WINO_MIN_PIX pixels per map (default 64 x 64) the dense blocks' gather convs run on the Winograd
    kernels (scaled split of the transformed gradients, the producer's epilogue backward in the 32-channel kernel's epilogue,
    hcf_conv_wino.h SC variants; transposed packs rebuilt on the device after optimiser steps). Same sums, another order: against
    the direct scaled kernel on the same steps (knob read at the start of each backward pass), with and without the fused epilogue
    backward, before and after an optimiser step; the Winograd form is reproducible run to run."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(29)
    hr = torch.rand(3, 3, 96, 160, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    if fused:
        monkeypatch.delenv("HCF_NO_EPI_FUSE", raising=False)
    else:
        monkeypatch.setenv("HCF_NO_EPI_FUSE", "1")
    res = {}
    for form, minpix in (("wino", "0"), ("direct", "1000000000")):
        monkeypatch.setenv("HCF_DGRAD_WINO_MIN_PIX", minpix)
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train().set_precision("f16x3")
        opt = torch.optim.SGD([q for q in net.parameters() if q.requires_grad], lr=1e-7)
        steps = []
        for it in range(2):
            for rep in range(2):                           # the same step twice: reproducible
                opt.zero_grad(set_to_none=True)
                _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
                nll.backward()
                steps.append((float(nll.detach()), spec_grads(net, cfg)))
            assert net.engine().fallback_count() == 0
            opt.step()
        res[form] = steps
    for k in (0, 2):
        for a, b in zip(res["wino"][k][1], res["wino"][k + 1][1]):
            assert np.array_equal(a, b)
    ndiff = 0
    for k in (0, 2):                                       # step 1 (host-built packs), step 2 (device refresh)
        (n0, g0), (n1, g1) = res["wino"][k], res["direct"][k]
        assert n0 == n1
        gmax = max(float(np.abs(x).max()) for x in g1)
        for a, b in zip(g0, g1):
            assert np.isfinite(a).all()
            ndiff += int(not np.array_equal(a, b))
            assert float(np.abs(a - b).max()) <= 2e-4 * max(float(np.abs(b).max()), 2e-5 * gmax)
    assert ndiff > 0                                       # the knob did select another kernel


def test_taped_forward_on_the_fat_schedule_equals_the_per_conv_schedule(monkeypatch):
    """The taped forward of a training pass runs its dense blocks as the inference pass does (run_rdb: conv 2j+1 + the old-input part
    of conv 2j+2 as one 64-wide launch, then the completion), and records one tape entry per conv: the same tensors in another
    summation order. Against HCF_NO_TAPE_FAT=1 (one launch per conv, read at the start of each training pass): nll, LR^ and every
    gradient, before and after an optimiser step (the fat packs are rebuilt on the device)."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(37)
    hr = torch.rand(3, 3, 96, 160, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    res = {}
    for form in ("fat", "per_conv"):
        if form == "fat":
            monkeypatch.delenv("HCF_NO_TAPE_FAT", raising=False)
        else:
            monkeypatch.setenv("HCF_NO_TAPE_FAT", "1")
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train().set_precision("f16x3")
        opt = torch.optim.SGD([q for q in net.parameters() if q.requires_grad], lr=1e-7)
        steps = []
        for it in range(2):
            opt.zero_grad(set_to_none=True)
            lr_hat, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
            nll.backward()
            steps.append((float(nll.detach()), lr_hat.detach().clone(), spec_grads(net, cfg)))
            assert net.engine().fallback_count() == 0
            opt.step()
        res[form] = steps
    ndiff = 0
    for it in range(2):
        (n0, l0, g0), (n1, l1, g1) = res["fat"][it], res["per_conv"][it]
        assert abs(n0 - n1) <= 1e-6 * abs(n1)
        assert float((l0 - l1).abs().max()) <= 1e-5
        gmax = max(float(np.abs(x).max()) for x in g1)
        for a, b in zip(g0, g1):
            assert np.isfinite(a).all()
            ndiff += int(not np.array_equal(a, b))
            assert float(np.abs(a - b).max()) <= 2e-4 * max(float(np.abs(b).max()), 2e-5 * gmax)
    assert ndiff > 0                                       # the knob did select another schedule


def test_training_step_on_a_side_stream_equals_the_default_stream():
    """The backward pass spreads over the caller's stream and the engine's own streams (weight gradients; the data gradients into the
    conditional features), tied together by events on whatever stream the caller is on: a step inside torch.cuda.stream(side) gives
    the gradients of the default-stream step bit for bit, with the optimiser's kernels queued right behind it on the same stream."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR, optim
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(41)
    hr = torch.rand(2, 3, 64, 96, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    res = []
    for use_side in (False, True):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train()
        opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-6)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ctx = torch.cuda.stream(side) if use_side else torch.cuda.stream(torch.cuda.current_stream())
        steps = []
        with ctx:
            for it in range(2):
                opt.zero_grad(set_to_none=True)
                _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
                nll.backward()
                grads = spec_grads(net, cfg)
                opt.step()
                steps.append((float(nll.detach()), grads))
        torch.cuda.synchronize()
        res.append(steps)
    for (n0, g0), (n1, g1) in zip(res[0], res[1]):
        assert n0 == n1
        for a, b in zip(g0, g1):
            assert np.array_equal(a, b)


def test_training_steps_between_split_inference_calls_keep_their_gradients():
    """The split inference calls and the training pass share the process' two side streams (hcf_aux_stream: half batches there,
    weight gradients / conditional-feature gradients here). Training steps with split sampling calls of another module in between
    (a validation loop inside a training loop, HCFlow_SR_model.py:209-262 between :184-205) give the gradients of the undisturbed
    steps bit for bit."""
    import numpy as np
    from hcflow_amd import HCFlowNet_SR, optim
    from hcflow_amd.config import preset
    from tests.util import cached_params, spec_grads
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(43)
    hr = torch.rand(4, 3, 64, 96, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    lr_val = torch.rand(6, 3, 12, 16, generator=g).cuda()

    def build(train):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0")
        return net.train() if train else net.eval()
    res, samples = [], []
    for disturb in (False, True):
        net, val = build(True), build(False).set_streams(2)
        opt = optim.Adam([q for q in net.parameters() if q.requires_grad], lr=1e-6)
        steps = []
        for it in range(3):
            if disturb:
                with torch.no_grad():
                    samples.append(val(lr=lr_val, eps_std=0.7, reverse=True, seed=5))     # split: two halves on the side streams
            opt.zero_grad(set_to_none=True)
            _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
            nll.backward()
            grads = spec_grads(net, cfg)
            opt.step()
            steps.append((float(nll.detach()), grads))
        torch.cuda.synchronize()
        res.append(steps)
        if disturb:
            assert len(val.engines()) == 2
    for (n0, g0), (n1, g1) in zip(res[0], res[1]):
        assert n0 == n1
        for a, b in zip(g0, g1):
            assert np.array_equal(a, b)
    assert all(torch.equal(samples[0], s) for s in samples[1:])


def test_two_phase_backward_equals_the_single_call_and_finishes_the_early_group_first():
    """hcf_train_backward_phase (DDP overlap): phase 0 + phase 1 write bit for bit what hcf_train_backward writes; after phase 0
    the slices of the parameters under flow.level0_condFlow. are already final (phase 1 does not touch them) and the rest is not;
    through the module, HCFLOW_GRAD_NODES=2 (two autograd nodes) gives the gradients of the one-node step."""
    import ctypes as C
    import os
    from hcflow_amd import _lib
    cfg, net = _fresh_sr("SR_4X_tiny", 11)
    net.train()
    g = torch.Generator().manual_seed(29)
    hr = torch.rand(2, 3, 64, 96, generator=g).cuda()
    lr = torch.rand(2, 3, 16, 24, generator=g).cuda()
    noise = torch.rand(hr.shape, generator=g).cuda()

    def step(nodes):
        os.environ["HCFLOW_GRAD_NODES"] = str(nodes)
        try:
            for p in net.parameters():
                p.grad = None
            _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
            nll.backward()
            return float(nll.detach()), [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        finally:
            os.environ.pop("HCFLOW_GRAD_NODES", None)
    n1, g1 = step(1)
    n2, g2 = step(2)
    assert n1 == n2
    assert all((a is None) == (b is None) for a, b in zip(g1, g2))
    assert all(torch.equal(a, b) for a, b in zip(g1, g2) if a is not None)
    # the C ABI directly: early slices final after phase 0
    eng = net.engine()
    lib, h = eng.lib, eng.handle
    keys = list(net._spec_keys)
    sizes = [p.numel() for p in net._params()]
    total = sum(sizes)
    flat = torch.zeros(total, device="cuda")
    out_lr = torch.empty(2, 3, 16, 24, device="cuda"); nll = torch.empty(1, device="cuda"); ld = torch.empty(2, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.hcf_train_forward_sr(h, hr.data_ptr(), lr.data_ptr(), noise.data_ptr(), out_lr.data_ptr(), nll.data_ptr(),
                                        ld.data_ptr(), 2, 64, 96, st), h, "fwd")
    _lib.check(lib.hcf_train_backward_phase(h, 0, 1.0, flat.data_ptr(), total, st), h, "phase 0")
    mid = flat.clone()
    assert lib.hcf_train_backward(h, 1.0, flat.data_ptr(), total, st) != 0          # phase 1 is pending: anything else is refused
    _lib.check(lib.hcf_train_backward_phase(h, 1, 1.0, flat.data_ptr(), total, st), h, "phase 1")
    off, early_n, late_changed = 0, 0, False
    for k, n in zip(keys, sizes):
        a, b = mid[off:off + n], flat[off:off + n]
        if k.startswith("flow.level0_condFlow."):
            assert torch.equal(a, b), k
            early_n += n
        else:
            late_changed = late_changed or not torch.equal(a, b)
        off += n
    assert 0 < early_n < total and late_changed
    _lib.check(lib.hcf_train_forward_sr(h, hr.data_ptr(), lr.data_ptr(), noise.data_ptr(), out_lr.data_ptr(), nll.data_ptr(),
                                        ld.data_ptr(), 2, 64, 96, st), h, "fwd")
    one = torch.zeros(total, device="cuda")
    _lib.check(lib.hcf_train_backward(h, 1.0, one.data_ptr(), total, st), h, "backward")
    assert torch.equal(one, flat)
    assert lib.hcf_train_backward_phase(h, 1, 1.0, one.data_ptr(), total, st) != 0    # no phase 0 before it
