"""CPU ORACLE for the HCFlow forward / inverse hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A functional restatement (PyTorch-CPU fp32 ops, no nn.Module, parameters passed as a flat
``{state_dict key: tensor}`` mapping) of the reference's algorithm for the path named by
BASELINE.json ``north_star``: HCFlowNet_SR / HCFlowNet_Rescaling ``forward(..., reverse=...)``.
Every function cites the reference file:line it follows. Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (``hcflow_amd``) never does and fails loudly when its HIP library is missing.

Pinning: the reference has no tests / golden vectors for this path, so the
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, generated in the build container by
``tests/golden/make_golden.py`` (which imports /root/reference) and committed as fixtures under
``tests/golden/`` -- see tests/test_oracle_golden.py.

Randomness: the reference draws ``torch.normal`` / ``torch.rand`` inside the pass
(Basic.py:96-100, HCFlowNet_SR_arch.py:52). For parity the draws are injectable: ``eps`` is the
list of N(0, tau) tensors in sampling order (deepest level first), ``noise`` the U[0,1) tensor.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

import re

from hcflow_amd.config import NetConfig      # the parsed options (yml values) only: the layer structure below is re-derived from the state dict

P = Dict[str, torch.Tensor]
LOG2PI = float(np.log(2 * np.pi))


# ------------------------------------------------------------------ index ops (bit-exact)
def squeeze2d(x: torch.Tensor) -> torch.Tensor:
    """Basic.squeeze2d, factor 2 (Basic.py:127-140): out[b, c*4+i*2+j, h, w] = x[b, c, 2h+i, 2w+j]."""
    B, C, H, W = x.shape
    assert H % 2 == 0 and W % 2 == 0, "{}".format((H, W, 2))
    out = x.new_empty(B, C * 4, H // 2, W // 2)
    for i in range(2):
        for j in range(2):
            out[:, i * 2 + j::4] = x[:, :, i::2, j::2]
    return out


def unsqueeze2d(x: torch.Tensor) -> torch.Tensor:
    """Basic.unsqueeze2d, factor 2 (Basic.py:143-157): inverse index map of squeeze2d."""
    B, C, H, W = x.shape
    assert C % 4 == 0, "{}".format(C)
    out = x.new_empty(B, C // 4, H * 2, W * 2)
    for i in range(2):
        for j in range(2):
            out[:, :, i::2, j::2] = x[:, i * 2 + j::4]
    return out


def split_half(x):
    """thops.split_feature(type="split") (thops.py:42-43): [:C//2], [C//2:]."""
    C = x.shape[1]
    return x[:, :C // 2], x[:, C // 2:]


def split_cross(x):
    """thops.split_feature(type="cross") (thops.py:44-45): even / odd channels."""
    return x[:, 0::2], x[:, 1::2]


def sum_chw(x: torch.Tensor) -> torch.Tensor:
    """thops.sum(dim=[1,2,3]) (thops.py:4-17): three successive single-dim fp32 sums."""
    return x.sum(dim=1, keepdim=True).sum(dim=2, keepdim=True).sum(dim=3, keepdim=True).flatten()


# ------------------------------------------------------------------ Haar (rescaling squeeze)
_HAAR_SIGNS = (
    ((1, 1), (1, 1)),      # k=0  LL
    ((1, -1), (1, -1)),    # k=1  weights[1,0,0,1] = weights[1,0,1,1] = -1   (Basic.py:457-458)
    ((1, 1), (-1, -1)),    # k=2  weights[2,0,1,0] = weights[2,0,1,1] = -1   (Basic.py:460-461)
    ((1, -1), (-1, 1)),    # k=3  weights[3,0,1,0] = weights[3,0,0,1] = -1   (Basic.py:463-464)
)


def haar_forward(x: torch.Tensor) -> torch.Tensor:
    """HaarDownsampling.forward, reverse=False (Basic.py:470-478).

    Grouped 2x2 stride-2 conv with +-1 weights, /4, output channel order k*C + c.
    """
    B, C, H, W = x.shape
    out = x.new_empty(B, 4 * C, H // 2, W // 2)
    for k in range(4):
        s = _HAAR_SIGNS[k]
        acc = None
        # accumulation order of the 2x2 window: (0,0), (0,1), (1,0), (1,1)
        for i in range(2):
            for j in range(2):
                t = x[:, :, i::2, j::2] * float(s[i][j])
                acc = t if acc is None else acc + t
        out[:, k * C:(k + 1) * C] = acc / 4.0
    return out


def haar_inverse(y: torch.Tensor) -> torch.Tensor:
    """HaarDownsampling.forward, reverse=True (Basic.py:479-487): conv_transpose2d with the same
    +-1 weights after the k*C+c -> c*4+k channel reshuffle; x[c,2h+i,2w+j] = sum_k s_k(i,j) y[k*C+c,h,w]."""
    B, C4, H, W = y.shape
    C = C4 // 4
    out = y.new_empty(B, C, H * 2, W * 2)
    for i in range(2):
        for j in range(2):
            acc = None
            for k in range(4):
                t = y[:, k * C:(k + 1) * C] * float(_HAAR_SIGNS[k][i][j])
                acc = t if acc is None else acc + t
            out[:, :, i::2, j::2] = acc
    return out


# ------------------------------------------------------------------ elementwise flow layers
def actnorm_forward(x, bias, logs):
    """ActNorm2d forward, reverse=False (ActNorms.py:45-53,58-64,87-90): (x + b) * exp(logs)."""
    return (x + bias) * torch.exp(logs)


def actnorm_inverse(x, bias, logs):
    """ActNorm2d forward, reverse=True (ActNorms.py:54,66,91-94): x * exp(-logs) - b."""
    return x * torch.exp(-logs) - bias


def actnorm_logdet(logs, pixels: int) -> torch.Tensor:
    """dlogdet = thops.sum(logs) * pixels, an fp32 0-dim tensor (ActNorms.py:72)."""
    return torch.sum(logs) * pixels


def actnorm_data_init(x: torch.Tensor, scale: float = 1.0):
    """ActNorm data-dependent init (ActNorms.py:37-43): returns (bias, logs)."""
    bias = -x.mean(dim=0, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
    var = ((x + bias) ** 2).mean(dim=0, keepdim=True).mean(dim=2, keepdim=True).mean(dim=3, keepdim=True)
    logs = torch.log(scale / (torch.sqrt(var) + 1e-6))
    return bias, logs


class InitParams(dict):
    """A parameter dict that also carries the set of ActNorm prefixes still awaiting their data-dependent
    initialisation (``module.inited == False`` in train() mode, ActNorms.py:29-43,78-80). The forward functions
    below fit and store bias / logs for those prefixes on the first tensor that reaches them, exactly where the
    reference's ``_ActNorm.forward`` does, and then use the fitted values."""

    def __init__(self, params, prefixes):
        super().__init__(params)
        self.pending = set(prefixes)


def maybe_actnorm_init(p, pre: str, x: torch.Tensor, scale: float = 1.0):
    """``if not self.inited: self.initialize_parameters(input)`` (ActNorms.py:78-80): a non-zero bias means
    "already trained" (:33-35), otherwise fit (:37-43)."""
    pending = getattr(p, "pending", None)
    if not pending or pre not in pending:
        return
    pending.discard(pre)
    if bool((p[pre + ".bias"] != 0).any()):
        return
    bias, logs = actnorm_data_init(x, scale)
    p[pre + ".bias"] = bias
    p[pre + ".logs"] = logs


def invconv_forward(x, W):
    """InvertibleConv1x1 forward (Permutations.py:70-71,99-100): z = conv2d(x, W[:, :, None, None])."""
    C = W.shape[0]
    return F.conv2d(x, W.view(C, C, 1, 1))


def invconv_logdet(W, pixels: int) -> torch.Tensor:
    """dlogdet = slogdet(W)[1] * pixels, evaluated on CPU in fp32 (Permutations.py:70)."""
    return torch.slogdet(W.to("cpu"))[1] * pixels


def invconv_inverse(x, W):
    """InvertibleConv1x1 reverse (Permutations.py:72-74,105): weight = inverse(W.double()).float()."""
    C = W.shape[0]
    Winv = torch.inverse(W.double()).float()
    return F.conv2d(x, Winv.view(C, C, 1, 1))


def invconv_lu_weight(p, pre: str, reverse: bool):
    """InvertibleConv1x1.get_weight, LU_decomposed branch (Permutations.py:78-92): returns (weight [C, C], sum(log_s)).
    l = self.l * l_mask + eye; u = self.u * l_mask^T + diag(sign_s * exp(log_s)); forward w = p @ (l @ u); reverse
    w = inverse(u.double()).float() @ (inverse(l.double()).float() @ p.inverse())."""
    l_, u_, log_s = p[pre + ".l"], p[pre + ".u"], p[pre + ".log_s"]
    perm, sign_s = p[pre + ".p"], p[pre + ".sign_s"]
    C = l_.shape[0]
    l_mask = torch.tril(torch.ones(C, C, dtype=l_.dtype), -1)
    eye = torch.eye(C, dtype=l_.dtype)
    l = l_ * l_mask + eye
    u = u_ * l_mask.transpose(0, 1).contiguous() + torch.diag(sign_s * torch.exp(log_s))
    if not reverse:
        w = torch.matmul(perm, torch.matmul(l, u))
    else:
        l = torch.inverse(l.double()).float()
        u = torch.inverse(u.double()).float()
        w = torch.matmul(u, torch.matmul(l, perm.inverse()))
    return w, torch.sum(log_s)


def logscale_of(scale):
    """FrEIA-style soft clamp (AffineCouplings.py:53,83): 0.318 * atan(2 * scale)."""
    return 0.318 * torch.atan(2 * scale)


# ------------------------------------------------------------------ conv sub-networks
def conv_actnorm(x, p: P, pre: str, pad: int):
    """Basic.Conv2d with do_actnorm=True (Basic.py:14-53): bias-free conv then ActNorm (no logdet)."""
    y = F.conv2d(x, p[pre + ".weight"], None, 1, pad)
    maybe_actnorm_init(p, pre + ".actnorm", y)
    return actnorm_forward(y, p[pre + ".actnorm.bias"], p[pre + ".actnorm.logs"])


def conv_zeros(x, p: P, pre: str):
    """Basic.Conv2dZeros (Basic.py:57-72): (conv3x3 + bias) * exp(logs * 3)."""
    y = F.conv2d(x, p[pre + ".weight"], p[pre + ".bias"], 1, 1)
    return y * torch.exp(p[pre + ".logs"] * 3)


def fcn(x, p: P, pre: str):
    """Basic.FCN.forward (Basic.py:441-447): relu(conv1 3x3+AN) -> relu(conv2 1x1+AN) -> Conv2dZeros 3x3."""
    x = F.relu(conv_actnorm(x, p, pre + ".conv1", 1))
    x = F.relu(conv_actnorm(x, p, pre + ".conv2", 0))
    return conv_zeros(x, p, pre + ".conv3")


def dense5(x, p: P, pre: str):
    """Shared body of DenseBlock / ResidualDenseBlock (Basic.py:349-356, 379-385):
    five 3x3 convs with dense concatenation, LeakyReLU(0.2) after the first four."""
    feats = [x]
    for i in range(1, 5):
        y = F.conv2d(torch.cat(feats, 1), p["%s.conv%d.weight" % (pre, i)], p["%s.conv%d.bias" % (pre, i)], 1, 1)
        feats.append(F.leaky_relu(y, 0.2))
    return F.conv2d(torch.cat(feats, 1), p[pre + ".conv5.weight"], p[pre + ".conv5.bias"], 1, 1)


def rdb(x, p: P, pre: str):
    """ResidualDenseBlock.forward (Basic.py:379-385): x5 * 0.2 + x."""
    return dense5(x, p, pre) * 0.2 + x


def rrdb(x, p: P, pre: str):
    """RRDB.forward (Basic.py:394-398): three RDBs, out * 0.2 + x."""
    out = rdb(x, p, pre + ".RDB1")
    out = rdb(out, p, pre + ".RDB2")
    out = rdb(out, p, pre + ".RDB3")
    return out * 0.2 + x


def coupling_net(x, p: P, pre: str, nn_module: str):
    return fcn(x, p, pre) if nn_module == "FCN" else dense5(x, p, pre)


# ------------------------------------------------------------------ couplings
def coupling(z, u, p: P, pre: str, kind: str, nn_module: str, lr_vs_others: bool, reverse: bool):
    """AffineCoupling (AffineCouplings.py:30-87) / AffineCoupling3shift (:118-160).

    Returns (z_out, sum_logscale per sample or None).
    """
    f = pre + ".f"
    if kind == "Affine":
        z1, z2 = split_half(z)
        h = coupling_net(z1 if u is None else torch.cat((z1, u), 1), p, f, nn_module)
        shift, scale = split_cross(h)
        ls = logscale_of(scale)
        if not reverse:
            z2 = (z2 + shift) * torch.exp(ls)
        else:
            z2 = z2 * torch.exp(-ls) - shift
        return torch.cat((z1, z2), 1), sum_chw(ls)
    assert kind == "Affine3shift"
    if lr_vs_others:
        z1, z2 = z[:, :3], z[:, 3:]
        h = coupling_net(z1 if u is None else torch.cat((z1, u), 1), p, f, nn_module)
        shift, scale = split_cross(h)
        ls = logscale_of(scale)
        if not reverse:
            z2 = (z2 + shift) * torch.exp(ls)
        else:
            z2 = z2 * torch.exp(-ls) - shift
        return torch.cat((z1, z2), 1), sum_chw(ls)
    z2, z1 = z[:, :3], z[:, 3:]
    if not reverse:
        shift = coupling_net(z1 if u is None else torch.cat((z1, u), 1), p, f, nn_module)
        z2 = z2 + shift
    else:
        shift = coupling_net(z1, p, f, nn_module)      # reference ignores u here (:152)
        z2 = z2 - shift
    return torch.cat((z2, z1), 1), None


def flowstep_forward(z, u, logdet, p: P, pre: str, perm: str, kind: str, nn_module: str,
                     lr_vs_others=True):
    """FlowStep.normal_flow (FlowStep.py:40-51). ``logdet`` is the running per-sample [B] tensor
    (or None, as the rescaling net passes) and is updated in the reference's order."""
    B, C, H, W = z.shape
    pix = H * W
    maybe_actnorm_init(p, pre + ".actnorm", z)
    z = actnorm_forward(z, p[pre + ".actnorm.bias"], p[pre + ".actnorm.logs"])
    if logdet is not None:
        logdet = logdet + actnorm_logdet(p[pre + ".actnorm.logs"], pix)
    if perm == "invconv" and (pre + ".permute.l") in p:      # LU_decomposed=True (Permutations.py:78-92)
        Wm, sum_log_s = invconv_lu_weight(p, pre + ".permute", reverse=False)
        z = invconv_forward(z, Wm)
        if logdet is not None:
            logdet = logdet + sum_log_s * pix
    elif perm == "invconv":
        Wm = p[pre + ".permute.weight"]
        z = invconv_forward(z, Wm)
        if logdet is not None:
            logdet = logdet + invconv_logdet(Wm, pix)
    z, sl = coupling(z, u, p, pre + ".affine", kind, nn_module, lr_vs_others, reverse=False)
    if sl is not None and logdet is not None:
        logdet = logdet + sl
    return z, logdet


def flowstep_inverse(z, u, p: P, pre: str, perm: str, kind: str, nn_module: str, lr_vs_others=True):
    """FlowStep.reverse_flow (FlowStep.py:53-64): coupling^-1, permute^-1, actnorm^-1."""
    z, _ = coupling(z, u, p, pre + ".affine", kind, nn_module, lr_vs_others, reverse=True)
    if perm == "invconv" and (pre + ".permute.l") in p:
        Wi, _ = invconv_lu_weight(p, pre + ".permute", reverse=True)
        z = invconv_forward(z, Wi)
    elif perm == "invconv":
        z = invconv_inverse(z, p[pre + ".permute.weight"])
    return actnorm_inverse(z, p[pre + ".actnorm.bias"], p[pre + ".actnorm.logs"])


# ------------------------------------------------------------------ Gaussian prior
def gaussian_logp(mean, logs, x):
    """GaussianDiag.logp (Basic.py:78-94): sum_chw(-0.5 (2 logs + (x-mean)^2 / exp(2 logs) + ln 2pi))."""
    ll = -0.5 * (logs * 2. + ((x - mean) ** 2) / torch.exp(logs * 2.) + LOG2PI)
    return sum_chw(ll)


def gaussian_sample(mean, logs, eps_std, eps=None):
    """GaussianDiag.sample (Basic.py:96-101): mean + exp(logs) * N(0, eps_std)."""
    if eps is None:
        eps = torch.normal(mean=torch.zeros_like(mean), std=torch.ones_like(logs) * eps_std)
    return mean + torch.exp(logs) * eps


def quantize(x):
    """Basic.Quant (Basic.py:186-196): forward round(clamp(x,0,1) * 255) / 255, backward the identity
    (straight-through, also outside [0, 1])."""
    q = (torch.clamp(x, 0, 1) * 255.).round() / 255.
    if x.requires_grad:
        return x + (q - x).detach()
    return q


# ------------------------------------------------------------------ conditional flow
def cond_features(u, p: P, pre: str, cfg: NetConfig):
    """ConditionalFlow.get_conditional_feature_SR / _Rescaling (ConditionalFlow.py:99-110)."""
    first = F.conv2d(u, p[pre + ".conv_first.weight"], p[pre + ".conv_first.bias"], 1, 1)
    f1 = first
    for n in range(cfg.rrdb_nb[0]):
        f1 = rrdb(f1, p, "%s.RRDB_trunk0.%d" % (pre, n))
    t = f1
    for n in range(cfg.rrdb_nb[1]):
        t = rrdb(t, p, "%s.RRDB_trunk1.%d" % (pre, n))
    f2 = F.conv2d(t, p[pre + ".trunk_conv1.weight"], p[pre + ".trunk_conv1.bias"], 1, 1) + first
    if cfg.sr:
        return torch.cat([f1, f2], 1)
    return f2


def condflow_forward(a, u, logdet, p: P, pre: str, cfg: NetConfig, level: int):
    """ConditionalFlow.forward, reverse=False (ConditionalFlow.py:46-57 SR, 70-82 rescaling).

    SR: returns (updated logdet incl. logp, cond feature). Rescaling: (z, cond feature).
    """
    cf = cond_features(u, p, pre, cfg)
    z = a
    for k in range(_plan_from_state(p, cfg)[1][level]):
        z, logdet = flowstep_forward(z, cf, logdet, p, "%s.additional_flow_steps.%d" % (pre, k),
                                     cfg.c_perm, cfg.c_coupling, cfg.c_nn_module)
    h = conv_zeros(cf, p, pre + ".f")
    mean, s = split_cross(h)
    if cfg.sr:
        logdet = logdet + gaussian_logp(mean, s, z)
        return logdet, cf
    ls = logscale_of(s)
    return (z - mean) * torch.exp(-ls), cf


def condflow_inverse(u, p: P, pre: str, cfg: NetConfig, level: int, eps_std, eps=None):
    """ConditionalFlow.forward, reverse=True (ConditionalFlow.py:59-69 SR, 84-96 rescaling)."""
    cf = cond_features(u, p, pre, cfg)
    h = conv_zeros(cf, p, pre + ".f")
    mean, s = split_cross(h)
    logs = s if cfg.sr else logscale_of(s)
    z = gaussian_sample(mean, logs, eps_std, eps)
    for k in reversed(range(_plan_from_state(p, cfg)[1][level])):
        z = flowstep_inverse(z, cf, p, "%s.additional_flow_steps.%d" % (pre, k),
                             cfg.c_perm, cfg.c_coupling, cfg.c_nn_module)
    return z, cf


# ------------------------------------------------------------------ FlowNet traversal
def _plan_from_state(p: P, cfg: NetConfig) -> Tuple[List[dict], List[int]]:
    """``flow.layers`` re-derived from the STATE DICT alone -- independently of the product's ``config.layer_plan`` -- the way the
    reference's constructors register their modules (FlowNet_SR_x4.py:33-64, FlowNet_SR_x8.py:33-70, FlowNet_Rescaling_x4.py:33-67):
    a level opens with a squeeze (parameter-free, or Haar with its frozen filters), the flow steps are the indices that own an
    ActNorm, a parameter-free Split closes the level; channel counts come from the tensors' shapes (ActNorm bias = C, the prior
    head's ``f.logs`` = 2 (C - n_split)); AffineCoupling3shift's mode from its DenseBlock's input width (3 = LR vs others,
    AffineCouplings.py:118-160). Returns (plan, additional flow steps per level)."""
    steps = {}
    for k, v in p.items():
        m = re.match(r"flow\.layers\.(\d+)\.actnorm\.bias$", k)
        if m:
            steps[int(m.group(1))] = int(v.shape[1])
    L = len({m.group(1) for m in (re.match(r"flow\.level(\d+)_condFlow\.", k) for k in p) if m})
    after = []
    for level in range(L):
        pre = "flow.level%d_condFlow.additional_flow_steps." % level
        after.append(len({k[len(pre):].split(".")[0] for k in p if k.startswith(pre)}))
    plan, idx, C = [], 0, 3
    for level in range(L):
        plan.append({"idx": idx, "type": "squeeze", "level": level, "C_in": C})
        idx += 1
        C *= 4
        while idx in steps:
            assert steps[idx] == C, ("flow step %d: ActNorm over %d channels at a %d-channel position" % (idx, steps[idx], C))
            w = p.get("flow.layers.%d.affine.f.conv1.weight" % idx)
            lrv = True if (cfg.sr or w is None) else (int(w.shape[1]) == 3)
            plan.append({"idx": idx, "type": "flowstep", "level": level, "C": C, "lr_vs_others": lrv})
            idx += 1
        ns = C - int(p["flow.level%d_condFlow.f.logs" % level].shape[0]) // 2
        plan.append({"idx": idx, "type": "split", "level": level, "C": C, "n_split": ns})
        idx += 1
        C = ns
    assert not any(i >= idx for i in steps), "flow steps beyond the last split"
    return plan, after


def _up(x, f):
    """F.interpolate(scale_factor=f, mode='nearest') (FlowNet_SR_x4.py:98,117)."""
    return F.interpolate(x, scale_factor=f, mode="nearest")


def flownet_forward(x, logdet, p: P, cfg: NetConfig):
    """FlowNet.normal_flow (FlowNet_SR_x4.py:84-101, FlowNet_SR_x8.py:91-116,
    FlowNet_Rescaling_x4.py:89-106).

    Returns (z_lr, logdet per sample, [fake_z per level, level order 0..L-1] for rescaling).
    """
    z = x
    ys: List[torch.Tensor] = []
    a_s: List[torch.Tensor] = []
    plan, _ = _plan_from_state(p, cfg)
    L = sum(1 for e in plan if e["type"] == "split")
    for ent in plan:
        pre = "flow.layers.%d" % ent["idx"]
        if ent["type"] == "squeeze":
            z = haar_forward(z) if cfg.squeeze == "haar" else squeeze2d(z)
        elif ent["type"] == "flowstep":
            z, logdet = flowstep_forward(z, None, logdet, p, pre, cfg.perm, cfg.coupling,
                                         cfg.nn_module, ent["lr_vs_others"])
        else:
            n = ent["n_split"]
            z, a = z[:, :n], z[:, n:]          # Basic.Split forward (Basic.py:495-497)
            ys.append(z)
            a_s.append(a)
    # hierarchical conditional prior, deepest level first (FlowNet_SR_x4.py:95-99, x8:104-114)
    cfs: Dict[int, torch.Tensor] = {}
    fake_z: Dict[int, torch.Tensor] = {}
    for level in reversed(range(L)):
        u = [ys[level]]
        for l2 in range(level + 1, L):
            u.append(_up(cfs[l2], 2 ** (l2 - level)))
        u = torch.cat(u, 1) if len(u) > 1 else u[0]
        r, cf = condflow_forward(a_s[level], u, logdet, p, "flow.level%d_condFlow" % level, cfg, level)
        cfs[level] = cf
        if cfg.sr:
            logdet = r
        else:
            fake_z[level] = r
    return z, logdet, [fake_z[l] for l in range(L)] if not cfg.sr else None


def flownet_inverse(z, p: P, cfg: NetConfig, eps_std, eps: Optional[Sequence[torch.Tensor]] = None):
    """FlowNet.reverse_flow (FlowNet_SR_x4.py:106-123, FlowNet_SR_x8.py:121-144,
    FlowNet_Rescaling_x4.py:111-128)."""
    cfs: Dict[int, torch.Tensor] = {}
    draw = 0
    plan, _ = _plan_from_state(p, cfg)
    L = sum(1 for e in plan if e["type"] == "split")
    for ent in reversed(plan):
        pre = "flow.layers.%d" % ent["idx"]
        if ent["type"] == "flowstep":
            z = flowstep_inverse(z, None, p, pre, cfg.perm, cfg.coupling, cfg.nn_module,
                                 ent["lr_vs_others"])
        elif ent["type"] == "squeeze":
            z = haar_inverse(z) if cfg.squeeze == "haar" else unsqueeze2d(z)
        else:
            level = ent["level"]
            u = [z]
            for l2 in range(level + 1, L):
                u.append(_up(cfs[l2], 2 ** (l2 - level)))
            u = torch.cat(u, 1) if len(u) > 1 else u[0]
            e = None if eps is None else eps[draw]
            draw += 1
            a, cf = condflow_inverse(u, p, "flow.level%d_condFlow" % level, cfg, level, eps_std, e)
            cfs[level] = cf
            z = torch.cat((z, a), 1)            # Basic.Split reverse (Basic.py:498-499)
    return z


# ------------------------------------------------------------------ top modules
def sr_forward(hr, lr, p: P, cfg: NetConfig, noise: Optional[torch.Tensor] = None):
    """HCFlowNet_SR.normal_flow_diracLR (HCFlowNet_SR_arch.py:47-67): returns (clamp(LR^), nll)."""
    B, C, H, W = hr.shape
    pixels = H * W
    if noise is None:
        noise = torch.rand(hr.shape)
    x = hr + noise / cfg.quant
    logdet = torch.zeros(B, dtype=hr.dtype) + float(-np.log(cfg.quant) * pixels)
    z, logdet, _ = flownet_forward(x, logdet, p, cfg)
    zq = quantize(z)
    objective = logdet + gaussian_logp(lr, -torch.ones_like(lr) * 6, zq)
    nll = ((-objective) / float(np.log(2.) * pixels)).mean()
    return torch.clamp(zq, 0, 1), nll


def sr_inverse(lr, p: P, cfg: NetConfig, eps_std, eps=None, clamp: bool = True):
    """HCFlowNet_SR.reverse_flow_diracLR (HCFlowNet_SR_arch.py:70-75)."""
    out = flownet_inverse(lr, p, cfg, eps_std, eps)
    return torch.clamp(out, 0, 1) if clamp else out


def rescale_forward(hr, p: P, cfg: NetConfig):
    """HCFlowNet_Rescaling.normal_flow_diracLR (HCFlowNet_Rescaling_arch.py:39-46):
    returns (clamp(LR^), fake_z1, fake_z2)."""
    z, _, fz = flownet_forward(hr, None, p, cfg)
    return torch.clamp(z, 0, 1), fz[0], fz[1]


def rescale_inverse(lr, p: P, cfg: NetConfig, eps_std, eps=None, clamp: bool = True):
    """HCFlowNet_Rescaling.reverse_flow_diracLR (HCFlowNet_Rescaling_arch.py:49-54)."""
    out = flownet_inverse(lr, p, cfg, eps_std, eps)
    return torch.clamp(out, 0, 1) if clamp else out


def draw_eps(cfg: NetConfig, B: int, h: int, w: int, eps_std: float, seed: int):
    """Deterministic N(0, eps_std) draws in sampling order, the way GaussianDiag.sample draws them
    (Basic.py:98-99) but from an explicit generator."""
    from hcflow_amd.config import eps_shapes
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(s, generator=g) * eps_std for s in eps_shapes(cfg, B, h, w)]
