"""-m gpu: engine-level behaviour added in round 2 -- global-sample Philox offsets for batch shards, kept conditional
features across tau / samples (BASELINE.json config 3), the asynchronous f16x3 range check with its exact re-run, a steady-state
pass that only enqueues (HIP-graph capture), and nn.DataParallel-style replicas."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _net(name, seed, precision, cls=None):
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    from hcflow_amd.config import preset
    from tests.util import cached_params
    cfg = preset(name)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    net.load_state_dict(cached_params(name, seed), strict=True)
    for m in net.modules():
        if "ActNorm" in type(m).__name__:
            m.inited = True
    return cfg, net.to("cuda:0").eval().set_precision(precision)


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["SR_4X_tiny", "SR_8X_tiny", "Rescaling_4X_tiny"])
def test_shards_with_sample_offsets_reproduce_the_full_batch(name, precision):
    """hcf_inverse_ex: shard r samples the eps of global samples [lo, hi): two half batches (and a 3 + 2 split) with the common
    seed equal the one-GPU batch bit for bit; without the offset the second shard would repeat the first shard's draws."""
    cfg, net = _net(name, 11, precision)
    g = torch.Generator().manual_seed(3)
    lr = torch.rand(5, 3, 12, 16, generator=g).cuda()
    with torch.no_grad():
        full = net(lr=lr, eps_std=0.8, reverse=True, seed=1234)
        parts = [net(lr=lr[lo:hi].contiguous(), eps_std=0.8, reverse=True, seed=1234, sample_offset=lo)
                 for lo, hi in ((0, 3), (3, 5))]
        assert torch.equal(torch.cat(parts, 0), full)
        naive = net(lr=lr[3:5].contiguous(), eps_std=0.8, reverse=True, seed=1234)
        assert not torch.equal(naive, full[3:5])


@pytest.mark.parametrize("precision", ["exact", "f16x3"])
@pytest.mark.parametrize("name", ["SR_8X_tiny", "SR_4X_tiny"])
def test_kept_conditional_features_are_bit_identical_over_a_tau_sweep(name, precision):
    """Config 3 (diverse sampling): the deepest level's conditional features + prior head depend on lr only
    (FlowNet_SR_x8.py:129): cache_cond=True keeps them across tau / samples; outputs equal the uncached ones bit for bit, and
    the cache is dropped when lr or a parameter changes."""
    cfg, net = _net(name, 12, precision)
    g = torch.Generator().manual_seed(4)
    lr = torch.rand(3, 3, 8, 12, generator=g).cuda()
    taus = [0.0, 0.3, 0.8, 0.8]
    with torch.no_grad():
        want = [net(lr=lr, eps_std=t, reverse=True, seed=50 + i) for i, t in enumerate(taus)]
        got = [net(lr=lr, eps_std=t, reverse=True, seed=50 + i, cache_cond=True) for i, t in enumerate(taus)]
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        # an in-place change of lr must not be served from the cache
        lr.mul_(0.5)
        a = net(lr=lr, eps_std=0.5, reverse=True, seed=9, cache_cond=True)
        b = net(lr=lr, eps_std=0.5, reverse=True, seed=9)
        assert torch.equal(a, b)
        # ... nor a parameter update
        for p in net.parameters():
            if p.dim() == 4 and p.shape[1] > 1 and float(p.abs().max()) > 0:
                p.mul_(1.01)
                break
        a = net(lr=lr, eps_std=0.5, reverse=True, seed=9, cache_cond=True)
        b = net(lr=lr, eps_std=0.5, reverse=True, seed=9)
        assert torch.equal(a, b)
        # a different pass in between (forward) invalidates the kept buffers inside the engine
        a1 = net(lr=lr, eps_std=0.5, reverse=True, seed=9, cache_cond=True)
        hr = torch.rand(3, 3, 8 * cfg.scale, 12 * cfg.scale, generator=g).cuda()
        net(hr=hr, lr=lr, reverse=False)
        a2 = net(lr=lr, eps_std=0.5, reverse=True, seed=9, cache_cond=True)
        assert torch.equal(a1, a2) and torch.equal(a2, b)


def test_range_overflow_is_rerun_exactly_in_sync_mode_and_reported_in_lazy_mode():
    """|activation| >= 65504 cannot be split into f16 hi / lo parts. sync (default): the module asks the engine after the pass
    and redoes it on the exact kernels; lazy: nothing waits, check_range() reports it."""
    cfg, net = _net("SR_4X_tiny", 11, "f16x3")
    g = torch.Generator().manual_seed(5)
    lr = (torch.rand(2, 3, 12, 12, generator=g) * 2e5).cuda()
    ok_lr = torch.rand(2, 3, 12, 12, generator=g).cuda()
    with torch.no_grad():
        n0 = net.engine().fallback_count()
        out = net(lr=lr, eps_std=0.5, reverse=True, seed=3)
        assert net.engine().fallback_count() == n0 + 1
        net.set_precision("exact")
        ref = net(lr=lr, eps_std=0.5, reverse=True, seed=3)
        net.set_precision("f16x3")
        assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
        # an in-range pass afterwards is not affected (the flag was cleared)
        a = net(lr=ok_lr, eps_std=0.5, reverse=True, seed=3)
        assert net.engine().fallback_count() == n0 + 1 and bool(torch.isfinite(a).all())
        net.set_range_check("lazy")
        assert net.check_range() is False
        net(lr=ok_lr, eps_std=0.5, reverse=True, seed=3)
        assert net.check_range() is False
        net(lr=lr, eps_std=0.5, reverse=True, seed=3)
        net(lr=ok_lr, eps_std=0.5, reverse=True, seed=3)          # the flag is sticky until somebody asks
        assert net.check_range() is True
        assert net.check_range() is False
        b = net(lr=ok_lr, eps_std=0.5, reverse=True, seed=3)
        assert torch.equal(a, b)
        net.set_range_check("sync")


@pytest.mark.parametrize("name", ["SR_4X_tiny", "Rescaling_4X_tiny"])
def test_two_stream_split_equals_the_single_stream_call(name):
    """set_streams(2) (the default): a call of >= 4 samples runs as two half batches on two side streams / two engines, joined before it
    returns. Every op is per-sample and the device draws are indexed by the global sample, so the outputs equal the one-stream
    call's bit for bit (tiny nets: same kernel schedule at every batch size) -- sampling path, injected eps, the rescaling
    forward, and the per-sample range fallback across the two halves."""
    from hcflow_amd.config import eps_shapes
    cfg, net = _net(name, 11, "f16x3")
    g = torch.Generator().manual_seed(8)
    lr = torch.rand(6, 3, 12, 16, generator=g).cuda()
    eps = [torch.randn(s, generator=g).cuda() * 0.6 for s in eps_shapes(cfg, 6, 12, 16)]
    with torch.no_grad():
        net.set_streams(1)
        one = [net(lr=lr, eps_std=0.6, reverse=True, seed=21), net(lr=lr, eps_std=0.6, reverse=True, eps=eps)]
        assert len(net.engines()) == 1
        net.set_streams(2)
        try:
            two = [net(lr=lr, eps_std=0.6, reverse=True, seed=21), net(lr=lr, eps_std=0.6, reverse=True, eps=eps)]
            assert len(net.engines()) == 2
            assert torch.equal(one[0], two[0]) and torch.equal(one[1], two[1])
            if not cfg.sr:
                hr = torch.rand(6, 3, 48, 64, generator=g).cuda()
                net.set_streams(1)
                a = net(hr=hr, reverse=False)
                net.set_streams(2)
                b = net(hr=hr, reverse=False)
                assert all(torch.equal(x, y) for x, y in zip(a, b))
            bad = lr.clone()
            bad[4, 0, 3, 3] = 9.0e4                           # second half (samples 3..5): flagged by the twin engine
            n0 = sum(e.fallback_count() for e in net.engines())
            out = net(lr=bad, eps_std=0.6, reverse=True, seed=21)
            assert sum(e.fallback_count() for e in net.engines()) == n0 + 1
            net.set_precision("exact")
            ex = net(lr=bad, eps_std=0.6, reverse=True, seed=21)
            net.set_precision("f16x3")
            assert torch.equal(out[4].view(torch.int32), ex[4].view(torch.int32))
            rest = [0, 1, 2, 3, 5]
            assert torch.equal(out[rest], two[0][rest])
        finally:
            net.set_streams(2)


@pytest.mark.parametrize("name", ["SR_4X_tiny", "Rescaling_4X_tiny"])
def test_deferred_parameter_check_redoes_a_call_whose_parameters_changed(name):
    """Eval-mode inference calls skip the ~1 ms parameter walk in front of their first launch once a check has found the parameters
    unchanged, and verify the stamp while the GPU runs the pass (arch.py: _engine_for(defer=True)). A parameter written in place
    between two calls must still be picked up: the stale pass is drained and the call redone on the checked path."""
    cfg, net = _net(name, 11, "f16x3")
    net.set_streams(2)                               # (the default; explicit so that the test also holds under HCFLOW_STREAMS=1)
    g = torch.Generator().manual_seed(9)
    lr = torch.rand(6, 3, 12, 16, generator=g).cuda()
    hr = torch.rand(6, 3, 48, 64, generator=g).cuda()

    def call():
        out = [net(lr=lr, eps_std=0.5, reverse=True, seed=3)]
        if not cfg.sr:
            out += list(net(hr=hr, reverse=False))
        return out

    def same(a, b):
        return all(torch.equal(x, y) for x, y in zip(a, b))
    with torch.no_grad():
        a = [call() for _ in range(3)]
        ents = list(net._engines.values())
        assert len(ents) == 2 and all(e.get("stable") for e in ents)         # (the split's twin engine included)
        assert same(a[0], a[1]) and same(a[1], a[2])
        for p in net.parameters():
            if p.dim() == 4 and p.shape[1] > 1 and float(p.abs().max()) > 0:
                p.mul_(1.05)
                break
        b = call()                                   # enqueued unchecked, found stale, redone
        assert not same(b, a[0])
        if cfg.sr:                                   # (the rescaling case's second call of `call()` is a checked, unchanged one already)
            assert not any(e.get("stable") for e in ents)
        c = call()                                   # checked: unchanged
        d = call()                                   # deferred again
        assert all(e.get("stable") for e in ents) and same(b, c) and same(c, d)
        net.invalidate()                             # the full host path packs the same weights
        e_ = call()
        assert all(float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max())) for x, y in zip(e_, b))


def test_split_with_and_without_the_helper_thread_gives_the_same_samples(monkeypatch):
    """HCF_SPLIT_THREADED=0 enqueues both halves of a split call from the calling thread, the default hands the second half to a
    helper thread: same launches on the same two streams, same outputs."""
    cfg, net = _net("SR_8X_tiny", 12, "f16x3")
    net.set_streams(2)
    g = torch.Generator().manual_seed(15)
    lr = torch.rand(7, 3, 8, 12, generator=g).cuda()
    with torch.no_grad():
        monkeypatch.setenv("HCF_SPLIT_THREADED", "1")
        a = [net(lr=lr, eps_std=0.6, reverse=True, seed=30 + i) for i in range(3)]
        monkeypatch.setenv("HCF_SPLIT_THREADED", "0")
        b = [net(lr=lr, eps_std=0.6, reverse=True, seed=30 + i) for i in range(3)]
        net.set_streams(1)
        c = [net(lr=lr, eps_std=0.6, reverse=True, seed=30 + i) for i in range(3)]
    assert all(torch.equal(x, y) and torch.equal(y, z) for x, y, z in zip(a, b, c))


def test_kept_conditional_features_with_the_two_stream_split():
    """cache_cond=True on a split call: each engine keeps ITS half's features; the split layout is part of the cache key, so a call
    that is split differently from the one that filled the caches (set_streams in between) refills them instead of reading the
    other layout's buffers. Outputs equal the uncached single-stream ones bit for bit (tiny net: one kernel schedule)."""
    cfg, net = _net("SR_8X_tiny", 12, "f16x3")
    g = torch.Generator().manual_seed(14)
    lr_a = torch.rand(6, 3, 8, 12, generator=g).cuda()
    lr_b = torch.rand(6, 3, 8, 12, generator=g).cuda()
    taus = [0.0, 0.4, 0.9]
    with torch.no_grad():
        net.set_streams(1)
        want_a = [net(lr=lr_a, eps_std=t, reverse=True, seed=70 + i) for i, t in enumerate(taus)]
        want_b = [net(lr=lr_b, eps_std=t, reverse=True, seed=70 + i) for i, t in enumerate(taus)]
        net.set_streams(2)
        try:
            got = [net(lr=lr_a, eps_std=t, reverse=True, seed=70 + i, cache_cond=True) for i, t in enumerate(taus)]
            assert len(net.engines()) == 2
            assert all(torch.equal(a, b) for a, b in zip(want_a, got))
            # the twin engine now holds lr_a's second half; an UNSPLIT cached call on lr_b, then a split one on lr_b
            net.set_streams(1)
            assert torch.equal(net(lr=lr_b, eps_std=taus[1], reverse=True, seed=71, cache_cond=True), want_b[1])
            net.set_streams(2)
            got_b = [net(lr=lr_b, eps_std=t, reverse=True, seed=70 + i, cache_cond=True) for i, t in enumerate(taus)]
            assert all(torch.equal(a, b) for a, b in zip(want_b, got_b))
        finally:
            net.set_streams(2)


def test_range_overflow_of_one_sample_reruns_that_sample_only():
    """hcf_check_range_samples: the range flag carries the sample whose tiles saw the overflow; the module (sync policy) redoes
    exactly that sample on the exact kernels (B = 1, its global sample index for the device draws) and leaves the others'
    f16x3 results alone -- with injected eps and with device draws."""
    from hcflow_amd.config import eps_shapes
    cfg, net = _net("SR_4X_tiny", 11, "f16x3")
    g = torch.Generator().manual_seed(6)
    clean = torch.rand(3, 3, 12, 20, generator=g)
    bad = clean.clone()
    bad[1, 2, 5, 7] = 9.0e4
    eps = [torch.randn(s, generator=g) * 0.5 for s in eps_shapes(cfg, 3, 12, 20)]
    with torch.no_grad():
        for kw in (dict(seed=3), dict(eps=eps)):
            ref_clean = net(lr=clean.cuda(), eps_std=0.5, reverse=True, **kw)
            n0 = net.engine().fallback_count()
            out = net(lr=bad.cuda(), eps_std=0.5, reverse=True, **kw)
            assert net.engine().fallback_count() == n0 + 1
            net.set_precision("exact")
            ex = net(lr=bad.cuda(), eps_std=0.5, reverse=True, **kw)
            net.set_precision("f16x3")
            assert torch.equal(out[1].view(torch.int32), ex[1].view(torch.int32))
            assert torch.equal(out[0], ref_clean[0]) and torch.equal(out[2], ref_clean[2])
            ok, slots = net.engine().check_range_samples()
            assert not ok and slots == 0


@pytest.mark.parametrize("batch", [2, 6])
@pytest.mark.parametrize("precision", ["exact", "f16x3"])
def test_steady_state_inverse_only_enqueues_and_can_be_graph_captured(precision, batch):
    """With the plan cached and the range flag read back asynchronously a steady-state call contains no allocation and no
    host-device synchronisation: the whole pass can be captured into a HIP graph and replayed (bit-identical to eager). batch 6:
    the call takes the two-stream split -- under capture both halves are enqueued by the capturing thread (fork / join by events:
    two branches of the graph), no helper thread."""
    cfg, net = _net("SR_4X_tiny", 11, precision)
    net.set_range_check("lazy").set_streams(2)
    g = torch.Generator().manual_seed(6)
    lr = torch.rand(batch, 3, 16, 16, generator=g).cuda()
    with torch.no_grad():
        eager = net(lr=lr, eps_std=0.8, reverse=True, seed=77)        # also sizes the workspace
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            net(lr=lr, eps_std=0.8, reverse=True, seed=77)            # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            out = net(lr=lr, eps_std=0.8, reverse=True, seed=77)
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        lr.copy_(torch.rand(batch, 3, 16, 16, generator=g))           # new LR contents, same graph
        graph.replay()
        torch.cuda.synchronize()
        again = net(lr=lr, eps_std=0.8, reverse=True, seed=77)
        assert torch.equal(out, again)
    assert net.check_range() is False


def test_dataparallel_replica_forward_and_backward():
    """What nn.DataParallel does on every forward (HCFlow_SR_model.py:33-36, non-distributed): replicate -> replicas whose
    parameters are plain tensor attributes -> forward -> backward through Broadcast into the wrapped module's .grad."""
    from torch.nn.parallel import replicate
    cfg, net = _net("SR_4X_tiny", 11, "exact")
    net.train()
    g = torch.Generator().manual_seed(8)
    hr = torch.rand(2, 3, 32, 48, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
    nll.backward()
    want = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    replica = replicate(net, [torch.device("cuda", 0)])[0]
    assert len(list(replica.parameters())) == 0
    _, nll2 = replica(hr=hr, lr=lr, reverse=False, noise=noise)
    nll2.backward()
    assert float(nll2.detach()) == float(nll.detach())
    for p, w in zip(net.parameters(), want):
        assert p.grad is not None and torch.equal(p.grad, w)
    with torch.no_grad():                                             # inference on a replica as well
        a = replica(lr=lr, eps_std=0.7, reverse=True, seed=5)
        b = net(lr=lr, eps_std=0.7, reverse=True, seed=5)
    assert torch.equal(a, b)
    wrapped = torch.nn.DataParallel(net, device_ids=[0])
    _, nll3 = wrapped(hr=hr, lr=lr, reverse=False, noise=noise)
    assert float(nll3.detach()) == float(nll.detach())


def test_cond_cache_is_not_served_to_a_new_tensor_at_a_recycled_address():
    """`for lr in loader: net(lr=lr.cuda(), cache_cond=True)` -- the caching allocator hands the freed LR batch's
    address (and _version 0) to the next same-shaped batch; the cache key holds its tensor, so the new batch recomputes."""
    cfg, net = _net("SR_4X_tiny", 12, "f16x3")
    g = torch.Generator().manual_seed(6)
    lrs = [torch.rand(2, 3, 8, 12, generator=g) for _ in range(4)]
    with torch.no_grad():
        wants = [net(lr=l.cuda(), eps_std=0.7, reverse=True, seed=40 + i) for i, l in enumerate(lrs)]      # uncached
        ptrs = []
        for i, l in enumerate(lrs):                       # nothing but cached calls in this loop: the key stays live
            lr = l.cuda()
            ptrs.append(lr.data_ptr())
            got = net(lr=lr, eps_std=0.7, reverse=True, seed=40 + i, cache_cond=True)
            assert torch.equal(got, wants[i]), i
            del lr, got
        # (the module keeps the keyed tensor alive, so the allocator CANNOT hand its address to the NEXT batch)
        assert all(a != b for a, b in zip(ptrs, ptrs[1:]))


def test_trunk_microbatch_knob_is_bit_identical(monkeypatch):
    """HCF_TRUNK_MB (the MALL experiment of profiles/r03_notes.md, off by default) only re-orders the RRDB trunk over sub-batches:
    every op is per sample, so the images must not change. The knob is read once per process: run the comparison in a child."""
    import subprocess
    import sys
    code = (
        "import sys, torch, contextlib\n"
        "sys.path.insert(0, '.')\n"
        "from hcflow_amd import HCFlowNet_SR, preset, make_params\n"
        "cfg = preset('SR_4X_tiny')\n"
        "with contextlib.redirect_stdout(sys.stderr):\n"
        "    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)\n"
        "net.load_state_dict(make_params(cfg, 11), strict=True)\n"
        "[setattr(m, 'inited', True) for m in net.modules() if 'ActNorm' in type(m).__name__]\n"
        "net = net.cuda().eval()\n"
        "lr = torch.rand(6, 3, 24, 40, generator=torch.Generator().manual_seed(2)).cuda()\n"
        "with torch.no_grad():\n"
        "    out = net(lr=lr, eps_std=0.8, reverse=True, seed=9)\n"
        "print('DIGEST %.10e %.10e' % (float(out.double().sum()), float((out.double() ** 2).sum())))\n")
    import os
    outs = []
    for mb in ("0", "4"):
        env = dict(os.environ, HCF_TRUNK_MB=mb)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert outs[0] == outs[1], outs


def test_reverse_walk_of_winograd_units_is_bit_identical(monkeypatch):
    """Round 5: consecutive Winograd launches enumerate their units in opposite directions (wino::Args.rev; MALL reuse between a
    dense block's convs). The order in which units are computed must not change a bit: the same call with every launch walking
    forward (HCF_WINO_REV=0, read per launch) and with the default alternation -- twice, so both parities of the launch counter
    meet every conv."""
    cfg, net = _net("SR_4X_tiny", 11, "f16x3")
    g = torch.Generator().manual_seed(12)
    lr = torch.rand(3, 3, 24, 40, generator=g).cuda()       # HR 96 x 160: ragged 8- and 16-row units, several per launch
    with torch.no_grad():
        monkeypatch.setenv("HCF_WINO_REV", "0")
        fwd = net(lr=lr, eps_std=0.8, reverse=True, seed=21)
        monkeypatch.delenv("HCF_WINO_REV", raising=False)
        eng = net.engine()
        eng.profile_convs(True)
        a = net(lr=lr, eps_std=0.8, reverse=True, seed=21)
        n_wino = eng.conv_time(9, 0, kind=4, reset=True)[1]
        eng.profile_convs(False)
        assert n_wino > 0                                    # the pass does run Winograd launches
        b = net(lr=lr, eps_std=0.8, reverse=True, seed=21)
        c = net(lr=lr[:1], eps_std=0.8, reverse=True, seed=21)      # an odd number of launches in between: the parity shifts
        d = net(lr=lr, eps_std=0.8, reverse=True, seed=21)
    assert torch.equal(a, fwd) and torch.equal(b, fwd) and torch.equal(d, fwd)
    assert torch.equal(c, fwd[:1])
