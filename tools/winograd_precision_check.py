"""Numerical check of a Winograd F(2x2, 3x3) formulation of the f16x3 convolutions: emulates it in the CPU oracle (input
transform in fp32, weights transformed in fp64 then split, three f16-operand products accumulated in fp32, output transform in
fp32) for every 3x3 conv of the full-depth nets and compares against an fp64 evaluation, next to plain fp32 and the direct f16x3
split.    python tools/winograd_precision_check.py
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import hcflow_oracle as O
from hcflow_amd.config import preset, eps_shapes
from hcflow_amd.params import make_params

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
S = 2048.0
MODE = {"m": "exact"}
_orig = F.conv2d


def wino(x, w, split_mode):
    B, C, H, W = x.shape
    K = w.shape[0]
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    # 4x4 patches at stride 2: [B, C, th, tw, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ai,bcyxij,nj->bcyxan", BT, d, BT)            # fp32 adds only (entries 0, +-1)
    U = torch.einsum("ai,kcij,nj->kcan", G, w.double(), G)         # fp64, then split
    if split_mode == "f32":
        M = torch.einsum("kcan,bcyxan->bkyxan", U.float(), V)
    else:
        Uh = U.float().half().float(); Ul = ((U - Uh.double()).float() * S).half().float() / S
        Vh = V.half().float(); Vl = (V - Vh).half().float()
        M = torch.einsum("kcan,bcyxan->bkyxan", Uh, Vh) + (torch.einsum("kcan,bcyxan->bkyxan", Ul, Vh) +
                                                            torch.einsum("kcan,bcyxan->bkyxan", Uh, Vl))
    Y = torch.einsum("pa,bkyxan,qn->bkyxpq", AT, M, AT)            # [B, K, th, tw, 2, 2]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K, Hp, Wp)
    return Y[:, :, :H, :W]


def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    w = w.to(x.dtype)
    if MODE["m"] == "exact" or w.shape[-1] != 3 or x.dtype != torch.float32:
        return _orig(x, w, b, stride, padding, *a, **k)
    if MODE["m"] == "f16x3":
        xh = x.half().float(); xl = (x - xh).half().float()
        wh = w.half().float(); wl = ((w - wh) * S).half().float() / S
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) + _orig(xl, wh, None, stride, padding))
    else:
        y = wino(x, w, "f32" if MODE["m"] == "wino_f32" else "f16x3")
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


F.conv2d = conv2d
torch.set_num_threads(8)
if __name__ == "__main__":
    # single-conv sanity
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 9, 11, generator=g); w = torch.randn(32, 64, 3, 3, generator=g) / 24
    ref = _orig(x.double(), w.double(), None, 1, 1)
    for m in ("exact", "f16x3", "wino_f32", "wino_f16x3"):
        MODE["m"] = m
        print("single conv %-10s max err vs fp64 %.2e" % (m, float((conv2d(x, w, None, 1, 1).double() - ref).abs().max())))
    for name, h in (("SR_DF2K_4X", 24), ("SR_CelebA_8X", 10), ("Rescaling_DF2K_4X", 24)):
        cfg = preset(name); p = make_params(cfg, 1234)
        p64 = {k: v.double() for k, v in p.items()}
        lr = torch.rand(2, 3, h, h, generator=g)
        eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 2, h, h)]
        inv = O.sr_inverse if cfg.sr else O.rescale_inverse
        with torch.no_grad():
            MODE["m"] = "exact"
            ref64 = inv(lr.double(), p64, cfg, 0.8, [e.double() for e in eps], clamp=False)
            res = {}
            for m in ("exact", "f16x3", "wino_f32", "wino_f16x3"):
                MODE["m"] = m
                res[m] = float((inv(lr, p, cfg, 0.8, eps, clamp=False).double() - ref64).abs().max())
        print(name, "scale %.2f" % float(ref64.abs().max()), " ".join("%s %.2e" % kv for kv in res.items()), flush=True)
