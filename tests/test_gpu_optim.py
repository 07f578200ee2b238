"""hcflow_amd.optim on the GPU: the one-launch Adam (C ABI hcf_adam_step) against torch.optim.Adam -- the optimiser the reference's
training caller builds (HCFlow_SR_model.py:118-120) -- on the same gradients, its state_dict interchange with torch.optim.Adam
(base_model.save_training_state / resume_training), the scheduler's clear_state, and a few training steps of a drop-in net where
the engine must pick the new parameter values up. Same arithmetic, other rounding: tolerances are relative to each tensor's max."""
from collections import defaultdict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (5, 1), (4096,), (4097,), (64, 32, 3, 3), (10000,), (1, 13, 1, 1), (8195,)]


def _twins(seed=0):
    g = torch.Generator().manual_seed(seed)
    a = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in SHAPES]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    return a, b


def _set_grads(a, b, seed, flat=True, skip=()):
    """The same random gradients on both sets; on `a` as slices of ONE flat buffer starting 1 float off a 16-byte boundary (what a
    drop-in net's backward leaves, unaligned slices included), on `b` as separate tensors."""
    g = torch.Generator().manual_seed(seed)
    total = sum(p.numel() for p in a)
    buf = (torch.randn(total + 1, generator=g) * 3).cuda()
    off = 1
    for i, (p, q) in enumerate(zip(a, b)):
        n = p.numel()
        if i in skip:
            p.grad = None
            q.grad = None
        else:
            sl = buf[off:off + n].view(p.shape)
            p.grad = sl if flat else sl.clone()
            q.grad = sl.clone()
        off += n


def _close(x, y, rel=2e-6):
    x, y = x.detach().double().cpu(), y.detach().double().cpu()
    scale = max(float(y.abs().max()), 1e-30)
    assert float((x - y).abs().max()) <= rel * scale, (float((x - y).abs().max()), scale)


@pytest.mark.parametrize("wd", [0.0, 0.05])
@pytest.mark.parametrize("flat", [True, False])
def test_adam_equals_torch_adam(wd, flat):
    from hcflow_amd import optim
    a, b = _twins()
    mine = optim.Adam(a, lr=2.5e-3, betas=(0.9, 0.99), weight_decay=wd)
    ref = torch.optim.Adam(b, lr=2.5e-3, betas=(0.9, 0.99), weight_decay=wd)
    versions = [p._version for p in a]
    for it in range(6):
        _set_grads(a, b, 100 + it, flat=flat)
        mine.step()
        ref.step()
    for p, q in zip(a, b):
        _close(p, q)
        _close(mine.state[p]["exp_avg"], ref.state[q]["exp_avg"])
        _close(mine.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"])
    assert all(p._version > v for p, v in zip(a, versions))          # the raw-pointer update is visible to version watchers
    assert len({p.untyped_storage().data_ptr() for p in a}) == 1       # one flat buffer


def test_adam_late_first_gradient_and_lr_schedule():
    """A tensor that gets its first gradient at step 3 starts its own step count (torch keeps state['step'] per tensor), and
    group['lr'] written by a scheduler takes effect."""
    from hcflow_amd import optim
    a, b = _twins(1)
    mine = optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99))
    for it in range(6):
        _set_grads(a, b, 200 + it, skip=(2, 5) if it < 2 else (7,) if it == 4 else ())
        for o in (mine, ref):
            o.param_groups[0]["lr"] = 1e-3 * (0.5 ** (it // 2))
            o.step()
    for p, q in zip(a, b):
        _close(p, q)
    sd = mine.state_dict()
    assert float(sd["state"][2]["step"]) == 4.0 and float(sd["state"][0]["step"]) == 6.0 and float(sd["state"][7]["step"]) == 5.0


def test_state_dict_interchange_with_torch_adam():
    from hcflow_amd import optim
    a, b = _twins(2)
    mine = optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99))
    for it in range(3):
        _set_grads(a, b, 300 + it)
        mine.step(); ref.step()
    # ours -> torch's, torch's -> ours, then both continue from the other's state
    a2, b2 = [torch.nn.Parameter(p.detach().clone()) for p in a], [torch.nn.Parameter(p.detach().clone()) for p in b]
    ref2 = torch.optim.Adam(b2, lr=7e-4)
    ref2.load_state_dict(mine.state_dict())
    mine2 = optim.Adam(a2, lr=7e-4)
    mine2.load_state_dict(ref.state_dict())
    assert mine2.param_groups[0]["lr"] == 1e-3 and ref2.param_groups[0]["betas"] == (0.9, 0.99)
    for it in range(3):
        _set_grads(a2, b2, 400 + it)
        mine2.step(); ref2.step()
        _set_grads(a, b, 400 + it)
        mine.step(); ref.step()
    for p, q, r, s in zip(a, b, a2, b2):
        _close(r, q)
        _close(s, p)
        _close(p, q)


def test_scheduler_clear_state_restarts_the_moments():
    """MultiStepLR_Restart(clear_state=True) replaces optimizer.state by an empty defaultdict (lr_scheduler.py)."""
    from hcflow_amd import optim
    a, b = _twins(3)
    mine = optim.Adam(a, lr=1e-3)
    ref = torch.optim.Adam(b, lr=1e-3)
    for it in range(4):
        if it == 2:
            mine.state = defaultdict(dict)
            ref.state = defaultdict(dict)
        _set_grads(a, b, 500 + it)
        mine.step(); ref.step()
    for p, q in zip(a, b):
        _close(p, q)
    assert float(mine.state_dict()["state"][0]["step"]) == 2.0


@pytest.mark.parametrize("norm_type", [2.0, float("inf"), 1.0])
def test_clip_grad_norm_and_value_equal_torch(norm_type):
    from hcflow_amd import optim
    a, b = _twins(4)
    _set_grads(a, b, 600, skip=(3,))                # a hole in the flat buffer: two runs
    n0 = optim.clip_grad_norm_(a, 5.0, norm_type=norm_type)
    n1 = torch.nn.utils.clip_grad_norm_(b, 5.0, norm_type=norm_type)
    assert abs(float(n0) - float(n1)) <= 1e-5 * float(n1)
    for p, q in zip(a, b):
        if q.grad is not None:
            _close(p.grad, q.grad, 1e-5)
    optim.clip_grad_value_(a, 0.01)
    torch.nn.utils.clip_grad_value_(b, 0.01)
    for p, q in zip(a, b):
        if q.grad is not None:
            _close(p.grad, q.grad, 1e-5)                 # (unclamped elements carry the two norm roundings)
            assert float(p.grad.abs().max()) <= 0.01
    # scattered gradients (separate allocations): torch's multi-tensor route, same result
    _set_grads(a, b, 601, flat=False)
    n0 = optim.clip_grad_norm_(a, 5.0, norm_type=norm_type)
    n1 = torch.nn.utils.clip_grad_norm_(b, 5.0, norm_type=norm_type)
    assert abs(float(n0) - float(n1)) <= 1e-5 * float(n1)


def test_training_steps_of_a_drop_in_net_with_the_native_optimiser():
    """Three NLL steps of the tiny SR net, once with hcflow_amd.optim (Adam + clip on the flat gradient) and once with torch's:
    the same losses and parameters (the engine re-binds to the flat parameter buffer and refreshes its packs after every step),
    and the whole gradient arrives as one run."""
    from hcflow_amd import HCFlowNet_SR, optim
    from hcflow_amd.config import preset
    from tests.util import cached_params
    cfg = preset("SR_4X_tiny")
    g = torch.Generator().manual_seed(7)
    hr = torch.rand(2, 3, 64, 96, generator=g).cuda()
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    noise = torch.rand(hr.shape, generator=g).cuda()
    runs = []
    for native in (True, False):
        net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
        net.load_state_dict(cached_params("SR_4X_tiny", 11), strict=True)
        for m in net.modules():
            if "ActNorm" in type(m).__name__:
                m.inited = True
        net = net.to("cuda:0").train()
        ps = [q for q in net.parameters() if q.requires_grad]
        opt = (optim.Adam if native else torch.optim.Adam)(ps, lr=1e-5, betas=(0.9, 0.99))
        clip = optim.clip_grad_norm_ if native else torch.nn.utils.clip_grad_norm_
        losses, norms = [], []
        for it in range(3):
            opt.zero_grad(set_to_none=True)
            _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
            nll.backward()
            if native:
                rr = optim._grad_runs(list(net.parameters()))
                assert rr is not None and len(rr) <= 2
            norms.append(float(clip(net.parameters(), 50.0)))
            opt.step()
            losses.append(float(nll.detach()))
        with torch.no_grad():
            _, nll = net(hr=hr, lr=lr, reverse=False, noise=noise)
        losses.append(float(nll))
        runs.append((losses, norms, [q.detach().cpu().numpy().copy() for q in ps]))
    (l0, n0, p0), (l1, n1, p1) = runs
    assert len(set(l0)) == len(l0)                               # the parameters did move between steps
    for x, y in zip(l0, l1):
        assert abs(x - y) <= 2e-5 * abs(y)
    for x, y in zip(n0, n1):
        assert abs(x - y) <= 1e-4 * abs(y)
    for x, y in zip(p0, p1):
        assert float(np.abs(x - y).max()) <= 1e-5 * max(float(np.abs(y).max()), 1e-3)


def test_optimizer_survives_a_copy_and_refuses_moved_parameters():
    """copy.deepcopy / pickle keep defaults, state and param_groups only (torch.optim.Optimizer.__getstate__): the flat buffers are
    rebuilt from what came along. A parameter that left the GPU after the optimiser was built is an error, not a silent copy back."""
    import copy
    from hcflow_amd import _lib, optim
    a, b = _twins(5)
    mine = optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99))
    for it in range(2):
        _set_grads(a, b, 700 + it)
        mine.step(); ref.step()
    twin = copy.deepcopy(mine)                       # its own parameter tensors (deep copies), moments and step counts
    a2 = twin.param_groups[0]["params"]
    for it in range(2):
        _set_grads(a2, b, 710 + it)
        twin.step(); ref.step()
    for p, q in zip(a2, b):
        _close(p, q)
    for p, q in zip(a, a2):
        assert not torch.equal(p, q)                 # the original did not move
    _set_grads(a, b, 720, skip=(3,))
    a[3].data = a[3].data.cpu()
    a[3].grad = torch.ones_like(a[3])
    with pytest.raises(_lib.HcfError):
        mine.step()


def test_alternating_gradient_buffers_and_late_tensors_reuse_the_right_tables():
    """The chunk tables are cached per gradient-pointer set: two flat buffers taken in turn, with one tensor skipping a step in
    between (its step count falls behind: another table layout for the same pointers), still equal torch.optim.Adam."""
    from hcflow_amd import optim
    a, b = _twins(6)
    mine = optim.Adam(a, lr=1e-3, betas=(0.9, 0.99))
    ref = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.99))
    total = sum(p.numel() for p in a)
    bufs = [torch.empty(total + 1, device="cuda") for _ in range(2)]
    g = torch.Generator().manual_seed(800)
    for it in range(8):
        buf = bufs[it & 1]
        buf.copy_((torch.randn(total + 1, generator=g) * 2).cuda())
        off = 1
        for i, (p, q) in enumerate(zip(a, b)):
            n = p.numel()
            skip = (it == 3 and i == 4)
            p.grad = None if skip else buf[off:off + n].view(p.shape)
            q.grad = None if skip else buf[off:off + n].view(p.shape).clone()
            off += n
        mine.step(); ref.step()
    for p, q in zip(a, b):
        _close(p, q)
    assert float(mine.state_dict()["state"][4]["step"]) == 7.0
