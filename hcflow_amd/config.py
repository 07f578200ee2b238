"""Network configuration + parameter specification for the HCFlow hot path.

The reference reads its architecture from the yml ``network_G`` block through
``opt_get`` (reference: codes/utils/util.py:1153-1161) inside
``FlowNet.__init__`` (codes/models/modules/FlowNet_SR_x4.py:17-27,
FlowNet_SR_x8.py, FlowNet_Rescaling_x4.py:15-29) and
``ConditionalFlow.__init__`` (codes/models/modules/ConditionalFlow.py:15-41).
``NetConfig.from_opt`` performs the same look-ups (same keys, same defaults) and
``param_spec`` enumerates the ``state_dict`` keys/shapes the reference modules
register, so that checkpoints written by the reference load strictly
(SURVEY.md section 8b "State-dict (strict)").
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple


def opt_get(opt, keys, default=None):
    """Nested dict lookup with default (semantics of reference utils/util.py:1153-1161)."""
    if opt is None:
        return default
    ret = opt
    for k in keys:
        ret = ret.get(k, None) if hasattr(ret, "get") else None
        if ret is None:
            return default
    return ret


@dataclass
class NetConfig:
    kind: str = "SR"                 # "SR" | "Rescaling"
    scale: int = 4
    in_nc: int = 3
    quant: float = 256.0
    L: int = 2
    K: List[int] = field(default_factory=lambda: [26, 26])
    after: List[int] = field(default_factory=lambda: [13, 13])   # splitOff.after_flowstep
    squeeze: str = "checkerboard"    # "checkerboard" | "haar"
    # main (unconditional) flow steps
    perm: str = "invconv"            # "invconv" | "none"
    coupling: str = "Affine"         # "Affine" | "Affine3shift"
    nn_module: str = "FCN"           # "FCN" | "DenseBlock"
    hidden: int = 64
    # conditional (splitOff) flow steps
    c_perm: str = "invconv"
    c_coupling: str = "Affine"
    c_nn_module: str = "FCN"
    c_hidden: int = 64
    rrdb_nb: Tuple[int, int] = (5, 5)
    rrdb_nf: int = 64
    rrdb_gc: int = 32
    # FlowStep(LU_decomposed=...) -> Permutations.InvertibleConv1x1(LU_decomposed=...) (FlowStep.py:9-10,20;
    # Permutations.py:41-57): W = P (L o mask + I) (U o mask^T + diag(sign_s exp(log_s))). No shipped yml sets it (FlowNet never
    # passes the argument); read from network_G.flowDownsampler.LU_decomposed. Applies to every invconv step of the net.
    lu: bool = False

    # ------------------------------------------------------------------ helpers
    @property
    def sr(self) -> bool:
        return self.kind == "SR"

    @property
    def n_feat_cond(self) -> int:
        """num_features_condition (ConditionalFlow.py:21)."""
        return 2 if self.sr else 1

    @property
    def cond_ch(self) -> int:
        return self.rrdb_nf * self.n_feat_cond

    def level_channels(self, level: int) -> int:
        """Channels after the squeeze of ``level`` (FlowNet_SR_x4.py:36)."""
        c = self.in_nc
        for l in range(level + 1):
            c = c * 4
            if l < level:
                c = c // 2 if l < self.L - 1 else 3
        return c

    def split_channels(self, level: int) -> int:
        """num_channels_split of the Split after ``level`` (FlowNet_SR_x4.py:51)."""
        return self.level_channels(level) // 2 if level < self.L - 1 else 3

    def num_levels_condition(self, level: int) -> int:
        """How many coarser cond-feature maps feed conv_first at ``level``.

        x4 / rescaling: level0 -> 1, level1 -> 0 (FlowNet_SR_x4.py:55,62;
        FlowNet_Rescaling_x4.py:62,69); x8: level0 -> 2, level1 -> 1, level2 -> 0
        (FlowNet_SR_x8.py:55,62,69).
        """
        return self.L - 1 - level

    @staticmethod
    def from_opt(opt: dict) -> "NetConfig":
        ng = opt["network_G"]
        which = ng.get("which_model_G", "HCFlowNet_SR")
        kind = "Rescaling" if "rescaling" in which.lower() else "SR"
        fd = ng["flowDownsampler"]
        L = opt_get(opt, ["network_G", "flowDownsampler", "L"])
        K = opt_get(opt, ["network_G", "flowDownsampler", "K"])
        if isinstance(K, int):
            K = [K] * (L + 1)
        after = opt_get(opt, ["network_G", "flowDownsampler", "splitOff", "after_flowstep"], 0)
        if isinstance(after, int):
            after = [after] * (L + 1)
        if not opt_get(opt, ["network_G", "flowDownsampler", "splitOff", "enable"], False):
            raise NotImplementedError("HCFlow configs always enable splitOff (hierarchical prior)")
        so = fd["splitOff"]
        if kind == "SR":
            quant = opt_get(opt, ["quant"], 256)
        else:
            quant = opt_get(opt, ["datasets", "train", "quant"], 256)
        scale = opt_get(opt, ["scale"])
        if kind == "SR" and scale not in (4, 8):
            raise NotImplementedError("Scale {} is not implemented".format(scale))
        rrdb_nb = opt_get(so, ["RRDB_nb"], [5, 5])
        cfg = NetConfig(
            kind=kind, scale=int(scale), in_nc=opt_get(opt, ["network_G", "in_nc"], 3),
            quant=float(quant), L=int(L), K=list(K), after=list(after),
            squeeze=opt_get(fd, ["squeeze"], "checkerboard") if kind == "Rescaling" else "checkerboard",
            perm=opt_get(fd, ["flow_permutation"], "invconv"),
            coupling=opt_get(fd, ["flow_coupling"], "Affine"),
            nn_module=opt_get(fd, ["nn_module"], "FCN"),
            hidden=opt_get(fd, ["hidden_channels"], 64),
            c_perm=so["flow_permutation"], c_coupling=so["flow_coupling"],
            c_nn_module=opt_get(so, ["nn_module"], "FCN"),
            c_hidden=opt_get(so, ["hidden_channels"], 64),
            rrdb_nb=(int(rrdb_nb[0]), int(rrdb_nb[1])),
            rrdb_nf=opt_get(so, ["RRDB_nf"], 64), rrdb_gc=opt_get(so, ["RRDB_gc"], 32),
            lu=bool(opt_get(fd, ["LU_decomposed"], False)),
        )
        cfg.validate()
        return cfg

    def validate(self):
        assert self.kind in ("SR", "Rescaling")
        assert self.L in (2, 3), "reference ships L=2 (x4, rescaling) and L=3 (x8)"
        assert (self.kind, self.L) in (("SR", 2), ("SR", 3), ("Rescaling", 2))
        assert 2 ** self.L == self.scale, "scale must equal 2**L"
        for l in range(self.L):
            assert 0 <= self.after[l] <= self.K[l]
        for p in (self.perm, self.c_perm):
            assert p in ("invconv", "none"), p
        for c in (self.coupling, self.c_coupling):
            assert c in ("Affine", "Affine3shift"), c
        for n in (self.nn_module, self.c_nn_module):
            assert n in ("FCN", "DenseBlock"), n
        assert self.squeeze in ("checkerboard", "haar")

    def to_opt(self) -> dict:
        """Inverse of from_opt: a minimal reference-style option dict."""
        which = "HCFlowNet_SR" if self.sr else "HCFlowNet_Rescaling"
        opt = {
            "scale": self.scale,
            "network_G": {
                "which_model_G": which, "in_nc": self.in_nc, "out_nc": self.in_nc,
                "flowDownsampler": {
                    "K": self.K[0] if len(set(self.K)) == 1 else list(self.K), "L": self.L,
                    "flow_permutation": self.perm, "flow_coupling": self.coupling,
                    "nn_module": self.nn_module, "hidden_channels": self.hidden,
                    "cond_channels": None,
                    "splitOff": {
                        "enable": True, "after_flowstep": list(self.after),
                        "flow_permutation": self.c_perm, "flow_coupling": self.c_coupling,
                        "nn_module": self.c_nn_module, "nn_module_last": "Conv2dZeros",
                        "hidden_channels": self.c_hidden,
                        "RRDB_nb": list(self.rrdb_nb), "RRDB_nf": self.rrdb_nf,
                        "RRDB_gc": self.rrdb_gc,
                    },
                },
            },
        }
        if self.lu:
            opt["network_G"]["flowDownsampler"]["LU_decomposed"] = True
        if self.sr:
            opt["quant"] = self.quant
        else:
            opt["network_G"]["flowDownsampler"]["squeeze"] = self.squeeze
            opt["datasets"] = {"train": {"quant": self.quant}}
        if isinstance(opt["network_G"]["flowDownsampler"]["K"], list):
            # reference indexes K[level]; a list of length L works as well as L+1
            pass
        return opt


# ---------------------------------------------------------------------------- presets
def preset(name: str) -> NetConfig:
    """Configs shipped by the reference (codes/options/test/*.yml network_G blocks). ``base@K=1,3,2;after=0,2;nb=0,2`` names a
    depth variant of ``base`` (flow steps per level, how many of them act on the split half, RRDB counts of the two trunks):
    the option space FlowNet.__init__ accepts beyond the shipped ymls (reference-generated fixtures: tests/golden/net_var_*)."""
    if "@" in name:
        base, mods = name.split("@", 1)
        c = preset(base)
        for kv in mods.split(";"):
            k, v = kv.split("=")
            vals = [int(x) for x in v.split(",")]
            if k == "K":
                c.K = vals
            elif k == "after":
                c.after = vals
            elif k == "nb":
                c.rrdb_nb = (vals[0], vals[1])
            else:
                raise KeyError(kv)
        c.validate()
        return c
    if name == "SR_DF2K_4X":        # test_SR_DF2K_4X_HCFlow.yml:52-76
        return NetConfig(kind="SR", scale=4, quant=64.0, L=2, K=[26, 26, 26], after=[13, 13],
                         rrdb_nb=(7, 7))
    if name == "SR_CelebA_8X":      # test_SR_CelebA_8X_HCFlow.yml:38-63
        return NetConfig(kind="SR", scale=8, quant=256.0, L=3, K=[26, 26, 26, 26],
                         after=[13, 13, 13], rrdb_nb=(5, 5))
    if name == "Rescaling_DF2K_4X":  # test_Rescaling_DF2K_4X_HCFlow.yml:50-77
        return NetConfig(kind="Rescaling", scale=4, quant=256.0, L=2, K=[14, 14, 14], after=[6, 6],
                         squeeze="haar", perm="none", coupling="Affine3shift",
                         nn_module="DenseBlock", hidden=32, rrdb_nb=(2, 1), rrdb_gc=16)
    # reduced-depth variants with the real channel widths: used by tests / golden fixtures
    if name == "SR_4X_tiny":
        return NetConfig(kind="SR", scale=4, quant=64.0, L=2, K=[4, 4, 4], after=[2, 2],
                         rrdb_nb=(1, 1))
    if name == "SR_8X_tiny":
        return NetConfig(kind="SR", scale=8, quant=256.0, L=3, K=[4, 4, 4, 4], after=[2, 2, 2],
                         rrdb_nb=(1, 1))
    if name == "Rescaling_4X_tiny":
        return NetConfig(kind="Rescaling", scale=4, quant=256.0, L=2, K=[5, 5, 5], after=[2, 2],
                         squeeze="haar", perm="none", coupling="Affine3shift",
                         nn_module="DenseBlock", hidden=32, rrdb_nb=(1, 1), rrdb_gc=16)
    # LU-decomposed invertible 1x1 convs (Permutations.py:41-57) in every flow step: no shipped yml selects them
    if name in ("SR_4X_tiny_LU", "SR_8X_tiny_LU", "Rescaling_4X_tiny_LU", "SR_DF2K_4X_LU"):
        c = preset(name[:-3])
        c.lu = True
        if c.perm == "none":           # the rescaling yml has no permutation in its main steps: give the LU variant one
            c.perm = "invconv"
        return c
    # narrow variants (RRDB_nf 8, hidden 8) for the checkpoint fixtures: a whole state dict in a few hundred KB
    if name == "SR_4X_micro":
        return NetConfig(kind="SR", scale=4, quant=64.0, L=2, K=[3, 3, 3], after=[1, 1], hidden=8, c_hidden=8,
                         rrdb_nb=(1, 1), rrdb_nf=8, rrdb_gc=4)
    if name == "Rescaling_4X_micro":
        return NetConfig(kind="Rescaling", scale=4, quant=256.0, L=2, K=[3, 3, 3], after=[1, 1],
                         squeeze="haar", perm="none", coupling="Affine3shift",
                         nn_module="DenseBlock", hidden=8, rrdb_nb=(1, 1), rrdb_nf=8, rrdb_gc=4)
    raise KeyError(name)


# ---------------------------------------------------------------------------- param spec
# kind tags drive the seeded parameter recipe (params.py) and the engine's packers
ParamSpec = Tuple[str, Tuple[int, ...], str]


def _conv(out: List[ParamSpec], p: str, cin: int, cout: int, k: int):
    out.append((p + ".weight", (cout, cin, k, k), "conv_w"))
    out.append((p + ".bias", (cout,), "conv_b"))


def _fcn(out: List[ParamSpec], p: str, cin: int, cout: int, hid: int):
    """Basic.FCN (Basic.py:426-447): Conv2d(+ActNorm) 3x3, Conv2d(+ActNorm) 1x1, Conv2dZeros 3x3."""
    out.append((p + ".conv1.weight", (hid, cin, 3, 3), "fcn_w"))
    out.append((p + ".conv1.actnorm.bias", (1, hid, 1, 1), "an_bias"))
    out.append((p + ".conv1.actnorm.logs", (1, hid, 1, 1), "an_logs"))
    out.append((p + ".conv2.weight", (hid, hid, 1, 1), "fcn_w"))
    out.append((p + ".conv2.actnorm.bias", (1, hid, 1, 1), "an_bias"))
    out.append((p + ".conv2.actnorm.logs", (1, hid, 1, 1), "an_logs"))
    out.append((p + ".conv3.weight", (cout, hid, 3, 3), "zeros_w"))
    out.append((p + ".conv3.bias", (cout,), "zeros_b"))
    out.append((p + ".conv3.logs", (cout, 1, 1), "zeros_logs"))


def _dense(out: List[ParamSpec], p: str, cin: int, cout: int, gc: int, last_kind="zeros_w"):
    """Basic.DenseBlock (Basic.py:329-356)."""
    for i in range(4):
        _conv(out, "%s.conv%d" % (p, i + 1), cin + i * gc, gc, 3)
    out.append((p + ".conv5.weight", (cout, cin + 4 * gc, 3, 3), last_kind))
    out.append((p + ".conv5.bias", (cout,), "zeros_b" if last_kind == "zeros_w" else "conv_b"))


def coupling_io(C: int, cond: int, coupling: str, lr_vs_others: bool) -> Tuple[int, int]:
    """(f_in_channels, f_out_channels) of the coupling network.

    AffineCoupling (AffineCouplings.py:18-19) / AffineCoupling3shift (:101-106).
    """
    if coupling == "Affine":
        return C // 2 + cond, (C - C // 2) * 2
    if lr_vs_others:
        return 3 + cond, (C - 3) * 2
    return C - 3 + cond, 3


def _flowstep(out: List[ParamSpec], p: str, C: int, cond: int, perm: str, coupling: str,
              nn_module: str, hid: int, lr_vs_others: bool = True, lu: bool = False):
    """FlowStep (FlowStep.py:8-38): actnorm, permute, affine."""
    out.append((p + ".actnorm.bias", (1, C, 1, 1), "an_bias"))
    out.append((p + ".actnorm.logs", (1, C, 1, 1), "an_logs"))
    if perm == "invconv" and lu:
        # Permutations.py:51-55: parameters l, log_s, u, then the buffers p, sign_s (state_dict lists a module's parameters
        # before its buffers); l_mask / eye are plain attributes and never enter the state dict
        out.append((p + ".permute.l", (C, C), "lu_l"))
        out.append((p + ".permute.log_s", (C,), "lu_log_s"))
        out.append((p + ".permute.u", (C, C), "lu_u"))
        out.append((p + ".permute.p", (C, C), "lu_p"))
        out.append((p + ".permute.sign_s", (C,), "lu_sign_s"))
    elif perm == "invconv":
        out.append((p + ".permute.weight", (C, C), "invconv"))
    fin, fout = coupling_io(C, cond, coupling, lr_vs_others)
    if nn_module == "FCN":
        _fcn(out, p + ".affine.f", fin, fout, hid)
    else:
        _dense(out, p + ".affine.f", fin, fout, hid)


def _rrdb_trunk(out: List[ParamSpec], p: str, nb: int, nf: int, gc: int):
    for n in range(nb):
        for r in (1, 2, 3):
            q = "%s.%d.RDB%d" % (p, n, r)
            for i in range(4):
                _conv(out, "%s.conv%d" % (q, i + 1), nf + i * gc, gc, 3)
            _conv(out, q + ".conv5", nf + 4 * gc, nf, 3)


def _condflow(out: List[ParamSpec], p: str, cfg: NetConfig, level: int):
    """ConditionalFlow (ConditionalFlow.py:15-41)."""
    C = cfg.level_channels(level)
    ns = cfg.split_channels(level)
    cin = ns + cfg.cond_ch * cfg.num_levels_condition(level)
    _conv(out, p + ".conv_first", cin, cfg.rrdb_nf, 3)
    _rrdb_trunk(out, p + ".RRDB_trunk0", cfg.rrdb_nb[0], cfg.rrdb_nf, cfg.rrdb_gc)
    _rrdb_trunk(out, p + ".RRDB_trunk1", cfg.rrdb_nb[1], cfg.rrdb_nf, cfg.rrdb_gc)
    _conv(out, p + ".trunk_conv1", cfg.rrdb_nf, cfg.rrdb_nf, 3)
    for k in range(cfg.after[level]):
        _flowstep(out, "%s.additional_flow_steps.%d" % (p, k), C - ns, cfg.cond_ch,
                  cfg.c_perm, cfg.c_coupling, cfg.c_nn_module, cfg.c_hidden, lu=cfg.lu)
    out.append((p + ".f.weight", ((C - ns) * 2, cfg.cond_ch, 3, 3), "zeros_w"))
    out.append((p + ".f.bias", ((C - ns) * 2,), "zeros_b"))
    out.append((p + ".f.logs", ((C - ns) * 2, 1, 1), "zeros_logs"))


def layer_plan(cfg: NetConfig) -> List[dict]:
    """The ``flow.layers`` ModuleList as built by FlowNet.__init__ (FlowNet_SR_x4.py:33-64)."""
    plan = []
    idx = 0
    C = cfg.in_nc
    for level in range(cfg.L):
        plan.append({"idx": idx, "type": "squeeze", "level": level, "C_in": C})
        idx += 1
        C = C * 4
        for k in range(cfg.K[level] - cfg.after[level]):
            plan.append({"idx": idx, "type": "flowstep", "level": level, "C": C,
                         "lr_vs_others": (k % 2 == 0) if not cfg.sr else True})
            idx += 1
        ns = cfg.split_channels(level)
        plan.append({"idx": idx, "type": "split", "level": level, "C": C, "n_split": ns})
        idx += 1
        C = ns
    return plan


def param_spec(cfg: NetConfig) -> List[ParamSpec]:
    """state_dict keys/shapes in the reference's registration order."""
    out: List[ParamSpec] = []
    plan = layer_plan(cfg)
    # ModuleList entries first (flow.layers.*), then level{l}_condFlow in creation order:
    # nn.Module registers children in assignment order: layers (created before the loop), then
    # level0_condFlow, level1_condFlow, ... as the loop reaches each level.
    for ent in plan:
        p = "flow.layers.%d" % ent["idx"]
        if ent["type"] == "squeeze" and cfg.squeeze == "haar":
            out.append((p + ".haar_weights", (4 * ent["C_in"], 1, 2, 2), "haar"))
        elif ent["type"] == "flowstep":
            _flowstep(out, p, ent["C"], 0, cfg.perm, cfg.coupling, cfg.nn_module, cfg.hidden,
                      ent["lr_vs_others"], lu=cfg.lu)
    for level in range(cfg.L):
        _condflow(out, "flow.level%d_condFlow" % level, cfg, level)
    return out


def param_count(cfg: NetConfig) -> int:
    n = 0
    for _, shape, _ in param_spec(cfg):
        m = 1
        for s in shape:
            m *= s
        n += m
    return n


def eps_shapes(cfg: NetConfig, B: int, h: int, w: int) -> List[Tuple[int, int, int, int]]:
    """Shapes of the Gaussian draws of one inverse pass, in sampling order (deepest level first).

    (h, w) is the LR size. SURVEY.md section 8a row a13.
    """
    out = []
    for level in reversed(range(cfg.L)):
        C = cfg.level_channels(level)
        ns = cfg.split_channels(level)
        f = 2 ** (cfg.L - 1 - level)
        out.append((B, C - ns, h * f, w * f))
    return out
