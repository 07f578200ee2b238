"""Batch-sharded multi-GPU sampling: one process per GPU, RCCL all-gather of the output batch only.

New functionality relative to the reference (its test loader is hard-wired to batch 1,
codes/data/__init__.py:24; only DDP training uses NCCL): every op of the path is per-sample
(SURVEY.md section 8e), so a batch is split into contiguous shards, each rank runs the whole net on
its shard with its own engine, and ONE ``all_gather_into_tensor`` assembles the result. With
``torch.distributed`` backend "nccl" this is RCCL over xGMI on MI355X; the same code runs over
"gloo" on CPU tensors (tests/test_dist_cpu.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.distributed as dist


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous, balanced shards: the first B % world ranks get one extra sample."""
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_apply(fn: Callable[..., torch.Tensor], x: torch.Tensor, group=None,
                  extras: Optional[Sequence[Optional[torch.Tensor]]] = None, with_offset: bool = False) -> torch.Tensor:
    """Run ``fn`` on this rank's batch shard of ``x`` (and of each tensor in ``extras``) and
    all-gather the outputs along dim 0. Every rank passes the same full ``x`` and gets the same full result.
    ``with_offset``: ``fn`` also receives ``offset=`` the index of the shard's first sample in the full batch."""
    if not (dist.is_available() and dist.is_initialized()):
        kw = {"offset": 0} if with_offset else {}
        return fn(x, **kw) if extras is None else fn(x, *extras, **kw)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = x.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    n_max = -(-B // world)
    if hi > lo:
        args = [x[lo:hi]]
        if extras is not None:
            args += [None if e is None else e[lo:hi] for e in extras]
        y = fn(*args, **({"offset": lo} if with_offset else {}))
    else:                                     # more ranks than samples: run one sample to learn the shape
        args = [x[:1]]
        if extras is not None:
            args += [None if e is None else e[:1] for e in extras]
        y = fn(*args, **({"offset": 0} if with_offset else {}))[:0]
    pad = torch.zeros((n_max,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    pad[: y.shape[0]] = y
    out = torch.empty((world * n_max,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)      # the only collective of the path
    if B == world * n_max:
        return out
    pieces = []
    for r in range(world):
        l, h = shard_bounds(B, world, r)
        pieces.append(out[r * n_max: r * n_max + (h - l)])
    return torch.cat(pieces, 0)


def common_seed(device, group=None, seed: Optional[int] = None) -> int:
    """One Philox seed for the whole sharded batch: rank 0's draw (it follows torch.manual_seed there), broadcast."""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    if not (dist.is_available() and dist.is_initialized()):
        return seed
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([seed], dtype=torch.int64, device=device if on_gpu else "cpu")
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return int(t.item())


def sharded_inverse(net, lr: torch.Tensor, eps_std: float, eps: Optional[Sequence[torch.Tensor]] = None, group=None,
                    seed: Optional[int] = None):
    """netG(lr=..., eps_std=..., reverse=True) over a batch sharded across the ranks of ``group``.

    ``eps`` (optional, parity runs): full-batch N(0, tau) tensors in sampling order; each rank uses its slice.
    Without ``eps`` the draws happen on the device: every rank uses the SAME seed and the offset of its shard in the batch, so
    sample k of the gathered batch gets the eps sample k would get on one GPU (hcf_inverse_ex) -- the reference seeds all ranks
    identically (train_HCFlow.py:43-46), which with shard-local indexing would repeat the same eps in every shard.
    """
    if eps is None:
        sd = common_seed(lr.device, group, seed)
        return sharded_apply(lambda s, offset: net(lr=s, z=None, u=None, eps_std=eps_std, reverse=True, seed=sd,
                                                   sample_offset=offset), lr, group, with_offset=True)
    return sharded_apply(lambda s, *e: net(lr=s, z=None, u=None, eps_std=eps_std, reverse=True, eps=list(e)),
                         lr, group, extras=list(eps))


def sharded_rescale_roundtrip(net, hr: torch.Tensor, eps_std: float = 1.0, group=None):
    """Config 4 of BASELINE.json: forward -> Quant -> inverse per shard (HCFlow_Rescaling_model.py:306-324),
    all-gather of the reconstructed HR only (LR^ stays local)."""
    def fn(s):
        lr_hat, _, _ = net(hr=s, reverse=False)
        lrq = (torch.clamp(lr_hat, 0, 1) * 255.).round() / 255.      # Basic.Quant.forward (Basic.py:187-191)
        return net(lr=lrq, z=None, u=None, eps_std=eps_std, reverse=True)
    return sharded_apply(fn, hr, group)


# ---------------------------------------------------------------- the benchmark's N-GPU step (bench.py --gpus N)
_pending = []        # (Work, input tensor) of the overlapped all-gathers still in flight


def gather_flush():
    """Wait for the overlapped all-gathers of gathered_step (RCCL: the current stream waits for the collective's stream)."""
    while _pending:
        w, _keep = _pending.pop(0)
        w.wait()


def gathered_step(net, lr: torch.Tensor, tau: float, seed: int, out_all: Optional[torch.Tensor] = None, group=None,
                  overlap: bool = False, **kw):
    """One bench step on this rank: sample THIS rank's shard (its LR batch; eps of global samples [rank B, (rank + 1) B) of the
    job-wide seed, hcf_inverse_ex) and, for N > 1, all-gather the output batch into ``out_all`` -- the only collective of the
    path (RCCL over xGMI with backend "nccl"). Weak scaling: per-rank work is fixed.
    ``overlap`` (default False: ``out_all`` is complete when the call returns; bench.py opts in): the all-gather is issued asynchronously on the collective's own stream behind this step's kernels and the NEXT
    step's sampling runs beside it (78.6 MB per rank and step at the bench size: 7 x that arrives per GPU at N = 8); the
    previous step's gather is waited for before ``out_all`` is written again, and timed_region / gather_flush() wait for the last."""
    on = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if on else 0
    out = net(lr=lr, z=None, u=None, eps_std=tau, reverse=True, seed=seed, sample_offset=rank * lr.shape[0], **kw)
    if out_all is not None and on and dist.get_world_size(group) > 1:
        gather_flush()
        src = out.contiguous()
        w = dist.all_gather_into_tensor(out_all, src, group=group, async_op=bool(overlap))
        if overlap:
            _pending.append((w, src))
    return out


def timed_region(step_fn: Callable[[int], object], steps: int, first: int = 0, group=None) -> float:
    """bench.py's timing contract: barrier + device synchronize, EXACTLY ``steps`` calls of ``step_fn(i)``, barrier +
    synchronize, wall time, MAX over ranks. Returns seconds (the same value on every rank)."""
    import time
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    cuda = torch.cuda.is_available()
    if on:
        dist.barrier(group=group)
    if cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(first + i)
    gather_flush()                                   # overlapped all-gathers of the last step: inside the timed window
    if on:
        dist.barrier(group=group)
    if cuda:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if on:
        gpu = cuda and dist.get_backend(group) == "nccl"
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        dt = float(tmax.item())
    return dt


# ---------------------------------------------------------------- config 5: the DDP NLL training step (bench.py --workload train)
def wrap_ddp(net, device: Optional[torch.device] = None, group=None):
    """The reference's wrap (HCFlow_SR_model.py:33-36): ``DistributedDataParallel(netG, device_ids=[torch.cuda.current_device()])``
    when a process group with more than one rank exists, the bare module otherwise (the reference's non-distributed branch is
    ``DataParallel`` over one device: the same single replica). CPU tensors (the gloo tests) take DDP's device_ids=None form."""
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not on:
        return net
    from torch.nn.parallel import DistributedDataParallel
    if device is not None and device.type == "cuda":
        return DistributedDataParallel(net, device_ids=[device.index if device.index is not None else torch.cuda.current_device()],
                                       process_group=group)
    return DistributedDataParallel(net, process_group=group)


def grad_allreduce_bytes(net) -> int:
    """Bytes DDP all-reduces per step: every trainable tensor's gradient, fp32 (config 5: 92.9 MB for the SR x4 net)."""
    mod = net.module if hasattr(net, "module") else net
    return int(sum(p.numel() * p.element_size() for p in mod.parameters() if p.requires_grad))


def train_step(net, hr: torch.Tensor, lr: torch.Tensor, optimizer, clip_fn=None, max_norm: float = 100.0, **fwd_kw):
    """One optimisation step exactly as the reference's caller runs it (HCFlow_SR_model.optimize_parameters :184-205 with the
    NLL loss only, gradient_clip :289-294): zero_grad, ``netG(hr=, lr=, reverse=False)`` by keyword (through DDP when wrapped:
    its hooks all-reduce the gradients during backward), ``nll.backward()``, ``clip_grad_norm_``, ``optimizer.step()``.
    Returns the detached loss tensor (no host sync here)."""
    optimizer.zero_grad(set_to_none=True)
    _, nll = net(hr=hr, lr=lr, reverse=False, **fwd_kw)
    nll.backward()
    mod = net.module if hasattr(net, "module") else net
    (clip_fn or torch.nn.utils.clip_grad_norm_)(mod.parameters(), max_norm)
    optimizer.step()
    return nll.detach()
