"""The optimiser step of the training caller on the GPU in one launch.

Reference: HCFlow_SR_model.py:118-120 (HCFlow_Rescaling_model.py:140-142) builds ``torch.optim.Adam(optim_params, lr=lr_G,
weight_decay=wd_G, betas=(beta1, beta2))`` over netG's ~1500 parameter tensors, ``gradient_clip`` (:289-294) calls
``torch.nn.utils.clip_grad_norm_`` / ``clip_grad_value_`` on them and ``optimize_parameters`` steps the optimiser once per
iteration (:202). With torch's optimiser that is a multi-tensor list walk per step (12-14 ms of host time beside a 70 ms
forward + backward on MI355X); here it is one HIP kernel over flat buffers (C ABI hcf_adam_step, csrc/hcf_optim.hip):

    optimizer_G = hcflow_amd.optim.Adam(optim_params, lr=..., weight_decay=..., betas=...)      # same arguments
    hcflow_amd.optim.clip_grad_norm_(netG.parameters(), max_grad_norm)                          # same arguments

``Adam`` is a ``torch.optim.Optimizer``: ``param_groups`` (the reference's lr schedulers write ``group['lr']``), ``zero_grad``,
``state`` being dropped by ``MultiStepLR_Restart(clear_state=True)`` (lr_scheduler.py) and ``state_dict()`` /
``load_state_dict()`` work as with ``torch.optim.Adam``, and checkpoints are interchangeable with it (``training_state`` files of
base_model.save_training_state). Same arithmetic as ``torch.optim.Adam`` (amsgrad=False), not the same rounding.

What it does to the parameters: each group's tensors are re-pointed (``p.data``) into ONE flat fp32 device buffer (64-float
aligned slots) when the optimiser is built, so ``nn.Parameter`` identities, shapes and values stay and only ``data_ptr()`` moves
(the drop-in nets re-bind their engine to the new addresses on the next call). Parameters must already be on the GPU (build the
optimiser after ``.to(device)``, as the reference does): there is no CPU path.
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib

_ALIGN = 64          # floats: every parameter slot starts on a 256-byte boundary
_CHUNK = 4096        # HCF_ADAM_CHUNK of include/hcflow.h

_chunk_dtype = np.dtype([("grad", np.uint64), ("offset", np.uint32), ("n", np.uint32)])      # struct hcf_adam_chunk


class _Flat:
    """One parameter group in flat form."""

    def __init__(self, params):
        if not params:
            raise ValueError("hcflow_amd.optim.Adam: empty parameter group")
        dev = params[0].device
        for p in params:
            if p.device.type != "cuda" or p.device != dev or p.dtype != torch.float32:
                raise _lib.HcfError("hcflow_amd.optim.Adam updates fp32 parameters of ONE GPU per group (got %s %s next to %s): "
                                    "build it after .to(device); there is no CPU path" % (p.device, p.dtype, dev))
        self.device = dev
        self.numel = [p.numel() for p in params]
        offs, o = [], 0
        for n in self.numel:
            offs.append(o)
            o += (n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offs, self.total = offs, max(o, _ALIGN)
        self.P = torch.zeros(self.total, device=dev, dtype=torch.float32)
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        for p, off, n in zip(params, offs, self.numel):
            self.adopt(p, off, n)
        self.pptr = [self.P.data_ptr() + 4 * off for off in offs]
        self.t = np.zeros(len(params), np.int64)         # Adam's step count per parameter (torch keeps state['step'] per tensor)
        self.key = None                                  # gradient pointers the current chunk tables were built for
        self.tables = []                                 # [(a member tensor's index, device chunk table, n_chunks, keep-alive)]
        self.recent = collections.OrderedDict()          # the last few (pointers -> tables); valid while the step-count classes stand

    def adopt(self, p, off, n):
        with torch.no_grad():
            slot = self.P[off:off + n].view(p.shape)
            slot.copy_(p.detach())
            p.data = slot

    def view(self, buf, i, p):
        return buf[self.offs[i]:self.offs[i] + self.numel[i]].view(p.shape)


class Adam(torch.optim.Optimizer):
    """``torch.optim.Adam`` (amsgrad=False, maximize=False) as one launch per parameter group and step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, foreach=None,
                 maximize=False, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False):
        if amsgrad:
            raise ValueError("hcflow_amd.optim.Adam: amsgrad is not implemented (the reference never sets it)")
        # torch.optim.Adam's keyword-only arguments are accepted at their defaults (foreach / fused only choose torch's own
        # implementation and are ignored: this class IS one fused launch); the ones that change the arithmetic are rejected by name
        for name, val in (("maximize", maximize), ("capturable", capturable), ("differentiable", differentiable),
                          ("decoupled_weight_decay", decoupled_weight_decay)):
            if val:
                raise ValueError("hcflow_amd.optim.Adam: %s=True is not implemented; use torch.optim.Adam for it" % name)
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: %r" % (lr,))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: %r" % (eps,))
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameters: %r" % (betas,))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: %r" % (weight_decay,))
        # the keys torch.optim.Adam keeps in a group, so either class loads the other's state_dict
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)
        self._ensure()

    def _ensure(self):
        """The flat buffers (built with the optimiser; rebuilt from the parameters' current values in a copy that lost them:
        ``torch.optim.Optimizer.__getstate__`` keeps defaults, state and param_groups only)."""
        if "_flat" not in self.__dict__:
            self._lib = _lib.load()
            self._flat = [_Flat(g["params"]) for g in self.param_groups]
            for g, fl in zip(self.param_groups, self._flat):      # moments that came along (unpickled state): into the flat buffers
                for i, p in enumerate(g["params"]):
                    st = self.state.get(p)
                    if st and "exp_avg" in st:
                        with torch.no_grad():
                            m, v = fl.view(fl.M, i, p), fl.view(fl.V, i, p)
                            m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
                        fl.t[i] = int(round(float(st.get("step", 0))))
                        self.state[p] = {"exp_avg": m, "exp_avg_sq": v}
        return self._flat

    def __getstate__(self):
        # what copy / pickle carry: torch's three entries, with each tensor's step count put back beside its moments (the live
        # state keeps the counts in one array per group, not as ~1 500 scalar tensors to bump per step)
        d = dict(super().__getstate__())
        state = {}
        for g, fl in zip(self.param_groups, self._ensure()):
            for i, p in enumerate(g["params"]):
                if p in self.state:
                    ent = dict(self.state[p])
                    ent["step"] = torch.tensor(float(fl.t[i]), dtype=torch.float32)
                    state[p] = ent
        d["state"] = state
        return d

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if "_flat" in self.__dict__:
            self._flat.append(_Flat(self.param_groups[-1]["params"]))

    # ---- state in torch.optim.Adam's format ---------------------------------------------------------------------------------
    def _init_state(self, fl, i, p):
        fl.view(fl.M, i, p).zero_()
        fl.view(fl.V, i, p).zero_()
        fl.t[i] = 0
        self.state[p] = {"exp_avg": fl.view(fl.M, i, p), "exp_avg_sq": fl.view(fl.V, i, p)}

    def state_dict(self):
        sd = super().state_dict()
        idx = 0
        for g, fl in zip(self.param_groups, self._ensure()):
            for i, p in enumerate(g["params"]):
                if idx in sd["state"]:
                    # copies: the live moments are slices of the flat buffers (a loader that keeps the tensors, as
                    # torch.optim.Optimizer.load_state_dict does for same-device values, must not end up inside them)
                    ent = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd["state"][idx].items()}
                    ent["step"] = torch.tensor(float(fl.t[i]), dtype=torch.float32)
                    sd["state"][idx] = ent
                idx += 1
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for g, fl in zip(self.param_groups, self._ensure()):
                if g.get("amsgrad"):
                    raise ValueError("hcflow_amd.optim.Adam: the loaded state was trained with amsgrad=True")
                for i, p in enumerate(g["params"]):
                    st = self.state.get(p)
                    if not st:
                        continue
                    m, v = fl.view(fl.M, i, p), fl.view(fl.V, i, p)
                    m.copy_(st["exp_avg"])
                    v.copy_(st["exp_avg_sq"])
                    fl.t[i] = int(round(float(st["step"])))
                    self.state[p] = {"exp_avg": m, "exp_avg_sq": v}
                fl.key = None
                fl.recent.clear()

    # ---- the step ---------------------------------------------------------------------------------------------------------
    def _tables(self, fl, params, ptrs, active):
        """Device chunk tables, one per distinct step count among the tensors that have a gradient (one, unless some tensor
        got its first gradient later than the others)."""
        for i in active:
            g = params[i].grad
            if g.device != fl.device or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != fl.numel[i] or g.is_sparse:
                raise _lib.HcfError("hcflow_amd.optim.Adam: gradient %d is not a dense contiguous fp32 tensor on %s" % (i, fl.device))
        act = np.asarray(active, np.int64)
        numel = np.asarray(fl.numel, np.int64)[act]
        offs = np.asarray(fl.offs, np.int64)[act]
        gp = np.asarray([ptrs[i] for i in active], np.uint64)
        tv = fl.t[act]
        tables = []
        for t in np.unique(tv):
            sel = tv == t
            n, o, g = numel[sel], offs[sel], gp[sel]
            nch = (n + _CHUNK - 1) // _CHUNK
            seg = np.repeat(np.arange(len(n)), nch)
            first = np.cumsum(nch) - nch
            k = np.arange(int(nch.sum())) - first[seg]                    # chunk index inside its tensor
            tab = np.empty(len(seg), _chunk_dtype)
            tab["grad"] = g[seg] + (4 * _CHUNK * k).astype(np.uint64)
            tab["offset"] = (o[seg] + _CHUNK * k).astype(np.uint32)
            tab["n"] = np.minimum(_CHUNK, n[seg] - _CHUNK * k).astype(np.uint32)
            host = torch.from_numpy(tab.view(np.uint8).reshape(-1))
            # synchronous upload: a cached table is launched against on WHATEVER stream is current at a later step() (a training
            # loop may move into torch.cuda.stream(side) after step 1), so the copy must not be ordered on one stream only
            # (tables are rebuilt only when the gradient pointers change)
            dev = host.to(fl.device, non_blocking=False)
            tables.append((int(act[sel][0]), dev, len(seg), host))
        return tables

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, fl in zip(self.param_groups, self._ensure()):
            params = group["params"]
            state = self.state
            ptrs, active = [0] * len(params), []
            for i, p in enumerate(params):
                g = p.grad
                if g is None:
                    continue
                if g.dtype is not torch.float32 or not g.is_contiguous() or g.numel() != fl.numel[i]:
                    # (checked on every step, not only when a chunk table is built: a table cached for these pointers must not be
                    #  replayed over a gradient that has since become strided / expanded / cast)
                    raise _lib.HcfError("hcflow_amd.optim.Adam: gradient %d is not a dense contiguous fp32 tensor" % i)
                ptrs[i] = g.data_ptr()
                active.append(i)
                if p.data_ptr() != fl.pptr[i]:                  # someone re-pointed p.data (a manual swap, a cast and back): take it back in
                    if p.device != fl.device or p.dtype != torch.float32 or p.numel() != fl.numel[i]:
                        raise _lib.HcfError("hcflow_amd.optim.Adam: parameter %d is now %s %s; it was fp32 on %s when the optimiser "
                                            "was built (build the optimiser after .to(device))" % (i, p.device, p.dtype, fl.device))
                    fl.adopt(p, fl.offs[i], fl.numel[i])
                if p not in state:                              # first gradient, or the scheduler cleared the state on a restart
                    self._init_state(fl, i, p)
                    fl.key = None
                    fl.recent.clear()
            if not active:
                continue
            key = tuple(ptrs)
            if fl.key != key:
                # (the caching allocator may hand the backward pass one of a few blocks in turn: the last few tables are kept
                #  instead of rebuilding and re-uploading ~1 500 records whenever the gradient buffer alternates)
                ta = fl.t[np.asarray(active, np.int64)]
                sig = (ta - ta[0]).tobytes()                    # which tensors share a step count (a table is per count)
                hit = fl.recent.get(key)
                if hit is None or hit[0] != sig:
                    hit = (sig, self._tables(fl, params, ptrs, active))
                    fl.recent[key] = hit
                    if len(fl.recent) > 4:
                        fl.recent.popitem(last=False)
                fl.tables, fl.key = hit[1], key
            beta1, beta2 = group["betas"]
            stream = torch.cuda.current_stream(fl.device).cuda_stream
            with torch.cuda.device(fl.device):
                for (first, dev, n, _keep) in fl.tables:          # one table per step count (`first`: a member tensor of it)
                    rc = self._lib.hcf_adam_step(fl.P.data_ptr(), fl.M.data_ptr(), fl.V.data_ptr(), dev.data_ptr(), n,
                                                 float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                                                 float(group["weight_decay"]), int(fl.t[first]) + 1, C.c_void_p(stream))
                    if rc != 0:
                        raise _lib.HcfError("hcf_adam_step failed (%d)" % rc)
            fl.t[active] += 1
            # the kernel wrote through raw pointers: tell autograd (and the nets' engines, which watch _version) the tensors changed
            torch.autograd.graph.increment_version([params[i] for i in active])
        return loss


_MAX_RUNS = 8


def _grad_runs(params):
    """The gradients of ``params`` as a few flat tensors: maximal runs of gradients lying back to back in one buffer (what the
    drop-in nets' backward leaves -- hcf_train_backward writes a single flat gradient, with holes only at frozen tensors).
    None when they are scattered (more than _MAX_RUNS runs): the callers then take torch's multi-tensor route."""
    gs = [p.grad for p in params if p.grad is not None]
    if not gs:
        return []
    runs, first, last, nxt = [], None, None, 0
    if gs[0].device.type != "cuda":
        raise _lib.HcfError("hcflow_amd.optim: gradients live on %s; this package runs on the GPU only" % (gs[0].device,))
    f32 = torch.float32
    for g in gs:                                         # (a ~1500-iteration host loop per step: two calls per tensor)
        if g.dtype != f32 or not g.is_contiguous():       # (an expanded / strided gradient inside a run: torch's route)
            return None
        ptr = g.data_ptr()
        if first is None or ptr != nxt:
            if first is not None:
                runs.append((first, last))
                if len(runs) >= _MAX_RUNS:
                    return None
            first = g
        last, nxt = g, ptr + 4 * g.numel()
    runs.append((first, last))
    out = []
    for a, b in runs:
        # the run's ends share one dense storage, so every address between them is fp32 gradient memory of this run (the callers
        # only apply order-free elementwise operations and norms to it)
        st = a.untyped_storage()
        if (b.untyped_storage().data_ptr() != st.data_ptr() or b.device != a.device or a.is_sparse or not a.is_contiguous()
                or not b.is_contiguous()):
            return None                                  # neighbours by address only
        total = b.storage_offset() + b.numel() - a.storage_offset()
        out.append(torch.empty(0, device=a.device, dtype=torch.float32).set_(st, a.storage_offset(), (total,), (1,)))
    return out


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """``torch.nn.utils.clip_grad_norm_`` (HCFlow_SR_model.gradient_clip :293-294) on the flat gradient: one norm, one scale."""
    params = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
    runs = _grad_runs(params)
    if runs is None:
        return torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type=norm_type, error_if_nonfinite=error_if_nonfinite,
                                              foreach=foreach)
    if not runs:
        return torch.tensor(0.0)
    norms = [torch.linalg.vector_norm(r, float(norm_type)) for r in runs]
    total = norms[0] if len(norms) == 1 else torch.linalg.vector_norm(torch.stack(norms), float(norm_type))
    if error_if_nonfinite and not bool(torch.isfinite(total)):
        raise RuntimeError("The total norm of order %s for gradients from `parameters` is non-finite, so it cannot be clipped"
                           % (norm_type,))
    coef = torch.clamp(float(max_norm) / (total + 1e-6), max=1.0)
    for r in runs:
        r.mul_(coef)
    return total


@torch.no_grad()
def clip_grad_value_(parameters, clip_value, foreach=None):
    """``torch.nn.utils.clip_grad_value_`` (HCFlow_SR_model.gradient_clip :291-292) on the flat gradient."""
    params = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
    runs = _grad_runs(params)
    if runs is None:
        return torch.nn.utils.clip_grad_value_(params, clip_value, foreach=foreach)
    for r in runs:
        r.clamp_(min=-float(clip_value), max=float(clip_value))
