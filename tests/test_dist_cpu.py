"""world_size-2 gloo test of the batch-sharding + output all-gather logic (runs on CPU).

The per-shard compute is the CPU oracle here (the HIP engine needs a GPU); what is under test is
hcflow_amd/dist.py: shard bounds, eps slicing, padding of uneven shards, the single collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hcflow_amd.dist import shard_bounds, sharded_apply, sharded_inverse


def test_shard_bounds_cover_batch():
    for B in (1, 2, 5, 16, 17, 64):
        for world in (1, 2, 3, 8):
            got = [shard_bounds(B, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == B
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in got]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import hcflow_oracle as O
        from hcflow_amd.config import preset, eps_shapes
        from hcflow_amd.params import make_params
        cfg = preset("SR_4X_tiny")
        p = make_params(cfg, 11)
        g = torch.Generator().manual_seed(0)
        lr = torch.rand(B, 3, 6, 8, generator=g)
        eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, B, 6, 8)]

        class Net:      # the drop-in class's call surface, oracle underneath
            def __call__(self, lr=None, z=None, u=None, eps_std=None, reverse=False, eps=None):
                with torch.no_grad():
                    return O.sr_inverse(lr, p, cfg, eps_std, eps)

        out = sharded_inverse(Net(), lr, 0.8, eps=eps)
        with torch.no_grad():
            ref = O.sr_inverse(lr, p, cfg, 0.8, eps)
        ok = out.shape == ref.shape and float((out - ref).abs().max()) <= 1e-5
        # device-sampling path: every shard is called with the common seed and ITS offset into the batch
        seen = []

        class Net2:
            def __call__(self, lr=None, z=None, u=None, eps_std=None, reverse=False, seed=None, sample_offset=0):
                seen.append((int(seed), int(sample_offset), int(lr.shape[0])))
                return lr * 2

        torch.manual_seed(100 + rank)                 # ranks disagree on their local RNG: rank 0's draw must win
        out2 = sharded_inverse(Net2(), lr, 0.8)
        ok = ok and torch.equal(out2, lr * 2)
        from hcflow_amd.dist import shard_bounds as sb
        lo, hi = sb(B, dist.get_world_size(), rank)
        seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(dist.get_world_size())]
        dist.all_gather(seeds, torch.tensor([seen[0][0]], dtype=torch.int64))
        ok = ok and len({int(t) for t in seeds}) == 1
        ok = ok and (seen[0][1:] == (lo, hi - lo) if hi > lo else seen[0][1:] == (0, 1))
        # generic path, shards see only their slice
        y = sharded_apply(lambda s: s * 2 + 1, torch.arange(B * 3, dtype=torch.float32).view(B, 3))
        ok = ok and torch.equal(y, torch.arange(B * 3, dtype=torch.float32).view(B, 3) * 2 + 1)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 3, 1])
def test_sharded_inverse_gloo_world2(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _an_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hcflow_amd.arch import HCFlowNet_SR, ActNorm2d
        from hcflow_amd.config import preset
        cfg = preset("SR_4X_tiny")
        torch.manual_seed(0)
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        an = [(k, m) for k, m in net.named_modules() if isinstance(m, ActNorm2d)]
        with torch.no_grad():
            for i, (_, m) in enumerate(an):                       # every rank "fitted" different values
                m.bias.fill_(1.0 + rank + 0.01 * i)
                m.logs.fill_(-0.5 * (rank + 1) + 0.01 * i)
        net._broadcast_actnorms(an)
        ok = all(float(m.bias.flatten()[0]) == pytest.approx(1.0 + 0.01 * i) and
                 float(m.logs.flatten()[0]) == pytest.approx(-0.5 + 0.01 * i) for i, (_, m) in enumerate(an))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_actnorm_init_is_broadcast_from_rank0_world2():
    """ActNorms.py:28-44 fits every rank's ActNorms to its own shard; the module broadcasts rank 0's fit so the DDP replicas
    start from one parameter set."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_an_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _rt_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from oracle import hcflow_oracle as O
        from hcflow_amd.config import preset
        from hcflow_amd.params import make_params
        from hcflow_amd.dist import sharded_rescale_roundtrip, gathered_step, timed_region, shard_bounds as sb
        cfg = preset("Rescaling_4X_tiny")
        p = make_params(cfg, 13)
        hr = torch.rand(B, 3, 16, 24, generator=torch.Generator().manual_seed(1))

        class Net:      # the rescaling class's call surface (forward -> (LR^, z1, z2); reverse -> HR), oracle underneath
            def __call__(self, hr=None, lr=None, z=None, u=None, eps_std=None, reverse=False):
                with torch.no_grad():
                    return O.rescale_inverse(lr, p, cfg, eps_std) if reverse else O.rescale_forward(hr, p, cfg)

        out = sharded_rescale_roundtrip(Net(), hr, eps_std=0.0)          # tau 0: deterministic decode
        with torch.no_grad():
            lr_hat, _, _ = O.rescale_forward(hr, p, cfg)
            ref = O.rescale_inverse((lr_hat.clamp(0, 1) * 255.).round() / 255., p, cfg, 0.0)
        ok = out.shape == ref.shape and float((out - ref).abs().max()) <= 1e-5
        # bench.py's N > 1 step: each rank samples ITS batch with the job seed and its global sample offset, ONE all-gather
        seen = []

        class Stub:
            def __call__(self, lr=None, z=None, u=None, eps_std=None, reverse=False, seed=None, sample_offset=0):
                seen.append((int(seed), int(sample_offset)))
                return lr.repeat_interleave(4, 2).repeat_interleave(4, 3) + float(seed)

        mine = torch.full((B, 3, 2, 2), float(rank))
        out_all = torch.empty(world * B, 3, 8, 8)

        def step(i):
            time.sleep(0.02 * (rank + 1))                                # rank 1 is the slow one
            gathered_step(Stub(), mine, 0.8, 4242 + i, out_all)

        dt = timed_region(step, 3, first=5)
        ok = ok and seen == [(4242 + 5 + i, rank * B) for i in range(3)]
        want = torch.cat([torch.full((B, 3, 8, 8), float(r) + 4242 + 7) for r in range(world)], 0)
        ok = ok and torch.equal(out_all, want)
        ts = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ts, torch.tensor([dt], dtype=torch.float64))
        ok = ok and float(ts[0]) == float(ts[1]) and dt >= 3 * 0.02 * world          # MAX over ranks, same on every rank
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [2, 3])
def test_sharded_rescale_roundtrip_and_bench_step_gloo_world2(B):
    """Config 4's multi-GPU leg (forward -> Quant -> inverse per shard, all-gather of the HR batch only) and the factored
    bench step / timing contract (hcflow_amd/dist.py: gathered_step, timed_region) on two gloo ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rt_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    assert sorted(res) == [(0, True), (1, True)]


class _StubSR(torch.nn.Module):
    """The SR class's training call surface (HCFlowNet_SR_arch.py:34-35: forward(hr=, lr=, reverse=False) -> (LR^, nll scalar)) on a
    two-layer stand-in, so that the config-5 harness logic runs on CPU."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = torch.nn.Conv2d(3, 4, 3, padding=1)
        self.b = torch.nn.Conv2d(4, 3, 1)
        self.frozen = torch.nn.Parameter(torch.ones(5), requires_grad=False)      # (the rescaling nets' Haar filters are frozen)

    def forward(self, hr=None, lr=None, z=None, u=None, eps_std=None, add_gt_noise=False, step=None, reverse=False, training=True):
        assert not reverse and hr is not None and lr is not None
        y = self.b(torch.relu(self.a(hr)))
        y = torch.nn.functional.avg_pool2d(y, 4)
        return y.detach(), ((y - lr) ** 2).mean()


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hcflow_amd.dist import wrap_ddp, train_step, timed_region, grad_allreduce_bytes
        net = _StubSR()
        ddp = wrap_ddp(net, torch.device("cpu"))
        ok = type(ddp).__name__ == "DistributedDataParallel" and ddp.module is net
        opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=1e-2, betas=(0.9, 0.99))
        g = torch.Generator().manual_seed(100 + rank)                   # every rank its own shard of the global batch
        hr = torch.rand(4, 3, 16, 16, generator=g)
        lr = torch.rand(4, 3, 4, 4, generator=g)
        losses = []

        def one(i):
            losses.append(float(train_step(ddp, hr, lr, opt, None, 100.0)))
        dt = timed_region(one, 3, first=0)
        ok = ok and len(losses) == 3 and losses[2] < losses[0] and dt > 0
        # DDP averaged the gradients: the replicas hold identical parameters after the steps although their data differ
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        ok = ok and torch.equal(both[0], both[1])
        ok = ok and grad_allreduce_bytes(ddp) == 4 * sum(p.numel() for p in net.parameters() if p.requires_grad)
        ok = ok and net.frozen.grad is None
        q.put((rank, bool(ok), losses[0]))
    finally:
        dist.destroy_process_group()


def test_config5_ddp_train_step_harness_gloo_world2():
    """bench.py --workload train on two gloo ranks with a stand-in net: the reference's wrap (HCFlow_SR_model.py:33-36), the
    step of optimize_parameters (:184-205, :289-294) and the timing contract (hcflow_amd/dist.py: wrap_ddp, train_step,
    timed_region); replicas stay in lock step on different shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)]
    assert res[0][2] != res[1][2]                    # different shards, different first losses


def test_wrap_ddp_without_a_process_group_is_the_module():
    from hcflow_amd.dist import wrap_ddp, grad_allreduce_bytes
    net = _StubSR()
    assert wrap_ddp(net, torch.device("cpu")) is net
    assert grad_allreduce_bytes(net) == 4 * sum(p.numel() for p in net.parameters() if p.requires_grad)


# ---------------------------------------------------------------- two autograd nodes: gradient all-reduce under the backward pass
class _TwoPhaseStub(torch.nn.Module):
    """A net whose training step goes through hcflow_amd.arch.two_phase_apply with a plain-torch state: `early` parameters get
    their gradients in phase 0, `late` ones in phase 1 -- the structure of the engine-backed NLL step (arch._SRNLLTwoPhase)."""
    def __init__(self, events):
        super().__init__()
        self.late = torch.nn.Linear(8, 8)               # registered first, as flow.layers.* is in the real net
        self.early = torch.nn.Linear(8, 8)
        self.events = events

    def forward(self, x):
        from hcflow_amd.arch import two_phase_apply
        net, ev = self, self.events

        class State:
            differentiable = (True,)

            def forward(self):
                with torch.enable_grad():
                    self.e = [p.detach().requires_grad_(True) for p in net.early.parameters()]
                    self.l = [p.detach().requires_grad_(True) for p in net.late.parameters()]
                    hmid = torch.tanh(torch.nn.functional.linear(x, self.l[0], self.l[1]))
                    self.loss = (torch.nn.functional.linear(hmid, self.e[0], self.e[1]) ** 2).mean()
                return (self.loss.detach(),)

            def backward(self, phase, g):
                if phase == 0:
                    self.g = g[0]
                    out = torch.autograd.grad(self.loss, self.e, self.g, retain_graph=True)
                    ev.append("phase0_done")
                    return list(out)
                ev.append("phase1_start")
                return list(torch.autograd.grad(self.loss, self.l, self.g))
        (loss,) = two_phase_apply(State(), list(self.early.parameters()), list(self.late.parameters()))
        return loss


def _two_phase_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torch.nn.parallel import DistributedDataParallel
        events = []
        torch.manual_seed(0)
        net = _TwoPhaseStub(events)
        ref = _TwoPhaseStub([])
        ref.load_state_dict(net.state_dict())
        ddp = DistributedDataParallel(net, bucket_cap_mb=1e-4)          # every parameter its own bucket

        def hook(state, bucket):
            last_iter = len(events) - 1 - events[::-1].index("iter")
            events.append(("bucket", bucket.index(), "phase1_start" in events[last_iter:]))
            fut = dist.all_reduce(bucket.buffer(), async_op=True).get_future()
            return fut.then(lambda f: f.value()[0] / world)
        ddp.register_comm_hook(None, hook)
        g = torch.Generator().manual_seed(50 + rank)
        x = torch.rand(4, 8, generator=g)
        overlapped = []
        for it in range(3):
            events.append("iter")
            for p in net.parameters():
                p.grad = None
            ddp(x).backward()
            cur = events[len(events) - 1 - events[::-1].index("iter"):]
            early_before = [e for e in cur if isinstance(e, tuple) and not e[2]]
            overlapped.append(len(early_before))
        # gradients = the plain autograd gradients of the same loss, averaged over the ranks
        xs = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(xs, x)
        want = [torch.zeros_like(p) for p in ref.parameters()]
        for xr in xs:
            hmid = torch.tanh(ref.late(xr))
            gr = torch.autograd.grad((ref.early(hmid) ** 2).mean(), list(ref.parameters()))
            want = [w + g_ / world for w, g_ in zip(want, gr)]
        ok = all(torch.allclose(p.grad, w, atol=1e-6) for p, w in zip(net.parameters(), want))
        q.put((rank, bool(ok), overlapped))
    finally:
        dist.destroy_process_group()


def test_two_autograd_nodes_let_ddp_reduce_the_first_buckets_before_the_backward_pass_ends_gloo_world2():
    """HCFlow_SR_model.py:33-36 wraps netG in DistributedDataParallel, whose bucketed all-reduce overlaps the backward pass. With the
    NLL step as two autograd nodes (hcflow_amd.arch.two_phase_apply; engine: hcf_train_backward_phase) the buckets of the `early`
    parameters are handed to the communication hook BEFORE phase 1 starts -- from the second iteration on (DDP rebuilds its buckets
    in gradient-arrival order after the first) -- and the reduced gradients are the plain autograd ones."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_phase_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p_ in procs:
        p_.join(60)
    assert sorted(r[:2] for r in res) == [(0, True), (1, True)]
    for _, _, overlapped in res:
        assert overlapped[1] >= 1 and overlapped[2] >= 1, overlapped      # buckets launched before phase 1, iterations 2 and 3
