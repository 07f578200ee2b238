"""Numerical check of Winograd formulations of the f16x3 convolutions: emulates them in the CPU oracle
(input transform V = Bt d B in fp32, weights U = G g Gt formed in fp64 then split, three f16-operand products accumulated in fp32,
output transform in fp32) for the 3x3 convs of the full-depth nets and compares against an fp64 evaluation, next to plain fp32 and
the direct f16x3 split.

Round 6: besides F(2x2,3x3) the table now has F(3x3,3x3) and F(4x4,3x3) (Cook-Toom matrices generated in
exact rationals from a point set, so alternative points can be tried), each under two policies:
  all    every 3x3 conv of the net takes the tile
  trunk  only the RRDB / dense-block convs (the 87 % of the FLOPs) take it, everything else stays F(2x2)
    python tools/winograd_precision_check.py [--size 24] [--nets SR_DF2K_4X,...] [--json out.json]
"""
import argparse, json, os, sys
from fractions import Fraction as Fr
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sympy
import torch, torch.nn.functional as F
from oracle import hcflow_oracle as O
from hcflow_amd.config import preset, eps_shapes
from hcflow_amd.params import make_params

S = 2048.0
_orig = F.conv2d


def cook_toom(m, pts, scale=None):
    """F(m, 3) over the finite points `pts` (+ infinity): returns (Bt [n,n], G [n,3], At [m,n]) as exact rationals.
    At[i][j] = a_j^i, G[j][k] = a_j^k / prod_{l != j}(a_j - a_l); Bt is the unique solution of
    sum_j At[i][j] G[j][k] Bt[j][t] = [t == i + k].  `scale[j]` moves a factor from G's row j into Bt's row j."""
    r = 3
    n = m + r - 1
    assert len(pts) == n - 1
    pts = [Fr(p) for p in pts]
    At = [[(pts[j] ** i if j < n - 1 else Fr(int(i == m - 1))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        f = Fr(1)
        for l in range(n - 1):
            if l != j:
                f *= pts[j] - pts[l]
        G.append([pts[j] ** k / f for k in range(r)])
    G.append([Fr(0), Fr(0), Fr(1)])
    Bt = [[None] * n for _ in range(n)]
    for t in range(n):
        rows, rhs = [], []
        for i in range(m):
            for k in range(r):
                rows.append([sympy.Rational(At[i][j] * G[j][k]) for j in range(n)])
                rhs.append(1 if t == i + k else 0)
        sol = sympy.linsolve((sympy.Matrix(rows), sympy.Matrix(rhs)))
        (vec,) = tuple(sol)
        assert not any(v.free_symbols for v in vec), "point set does not determine Bt"
        for j in range(n):
            Bt[j][t] = Fr(int(vec[j].p), int(vec[j].q))
    if scale is not None:
        for j in range(n):
            s = Fr(scale[j])
            G[j] = [g / s for g in G[j]]
            Bt[j] = [b * s for b in Bt[j]]
    return Bt, G, At


def _t(M, dt):
    return torch.tensor([[float(v) for v in row] for row in M], dtype=dt)


class Tile:
    def __init__(self, name, m, pts, scale=None):
        self.name, self.m = name, m
        self.n = m + 2
        Bt, G, At = cook_toom(m, pts, scale)
        self.Bt, self.G, self.At = _t(Bt, torch.float32), _t(G, torch.float64), _t(At, torch.float32)
        self.amp = float(self.Bt.abs().sum(1).max()) ** 2          # worst-case |V| / max|d|


TILES = {
    "F2": Tile("F(2x2,3x3)", 2, [0, 1, -1], scale=[1, 2, 2, 1]),           # the shipped kernels' matrices (G rows 1/2)
    "F3": Tile("F(3x3,3x3) pts 0,+-1,2", 3, [0, 1, -1, 2]),
    "F3h": Tile("F(3x3,3x3) pts 0,+-1,1/2", 3, [0, 1, -1, Fr(1, 2)]),
    "F4": Tile("F(4x4,3x3) pts 0,+-1,+-2", 4, [0, 1, -1, 2, -2]),
    "F4h": Tile("F(4x4,3x3) pts 0,+-1,+-1/2", 4, [0, 1, -1, Fr(1, 2), Fr(-1, 2)]),
}


def wino(x, w, tile, split_mode):
    B, C, H, W = x.shape
    m, n = tile.m, tile.n
    Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, n, m).unfold(3, n, m)                              # [B, C, th, tw, n, n]
    V = torch.einsum("ai,bcyxij->bcyxaj", tile.Bt, d)                   # fp32, rows then columns as a kernel would
    V = torch.einsum("bcyxaj,nj->bcyxan", V, tile.Bt)
    U = torch.einsum("ai,kcij,nj->kcan", tile.G, w.double(), tile.G)   # fp64, then split
    if split_mode == "f32":
        M = torch.einsum("kcan,bcyxan->bkyxan", U.float(), V)
    else:
        Uh = U.float().half().float(); Ul = ((U - Uh.double()).float() * S).half().float() / S
        Vh = V.half().float(); Vl = (V - Vh).half().float()
        M = torch.einsum("kcan,bcyxan->bkyxan", Uh, Vh) + (torch.einsum("kcan,bcyxan->bkyxan", Ul, Vh) +
                                                            torch.einsum("kcan,bcyxan->bkyxan", Uh, Vl))
    Y = torch.einsum("pa,bkyxan->bkyxpn", tile.At, M)
    Y = torch.einsum("bkyxpn,qn->bkyxpq", Y, tile.At)                   # [B, K, th, tw, m, m]
    Y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B, K_(Y), Hp, Wp)
    return Y[:, :, :H, :W]


def K_(Y):
    return Y.shape[1]


MODE = {"m": "exact", "tile": "F2", "policy": "all", "maxV": 0.0}


def is_trunk(w):
    """RRDB / dense-block growth convs: the layers conv_wino4 / conv_wino2 carry in the product (Basic.py:329-398)."""
    co, ci = w.shape[0], w.shape[1]
    return ci >= 64 and ci % 16 == 0 and co in (16, 32, 64) and ci in (64, 80, 96, 112, 128, 160, 192)


def conv2d(x, w, b=None, stride=1, padding=0, *a, **k):
    w = w.to(x.dtype)
    m = MODE["m"]
    if m == "exact" or w.shape[-1] != 3 or x.dtype != torch.float32:
        return _orig(x, w, b, stride, padding, *a, **k)
    if m == "f16x3":
        xh = x.half().float(); xl = (x - xh).half().float()
        wh = w.half().float(); wl = ((w - wh) * S).half().float() / S
        y = _orig(xh, wh, None, stride, padding) + (_orig(xh, wl, None, stride, padding) + _orig(xl, wh, None, stride, padding))
    else:
        tile = TILES[MODE["tile"]]
        if MODE["policy"] == "trunk" and not is_trunk(w):
            tile = TILES["F2"]
        y = wino(x, w, tile, "f32" if m == "wino_f32" else "f16x3")
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


F.conv2d = conv2d
torch.set_num_threads(8)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=24)
    ap.add_argument("--nets", default="SR_DF2K_4X,SR_CelebA_8X,Rescaling_DF2K_4X")
    ap.add_argument("--tiles", default="F2,F3,F3h,F4,F4h")
    ap.add_argument("--seeds", default="1234")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    tiles = a.tiles.split(",")
    rec = {"single_conv": {}, "nets": {}, "tiles": {t: {"name": TILES[t].name, "V_amplification": TILES[t].amp} for t in tiles}}
    for t in tiles:
        print("%-4s %-28s |V| <= %.0f x max|d|" % (t, TILES[t].name, TILES[t].amp))
    # single-conv sanity (also proves the generated matrices: wino_f32 must agree with the direct conv to fp32 noise)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 64, 13, 11, generator=g); w = torch.randn(32, 64, 3, 3, generator=g) / 24
    ref = _orig(x.double(), w.double(), None, 1, 1)
    for m, t in [("exact", "-"), ("f16x3", "-")] + [(mm, t) for t in tiles for mm in ("wino_f32", "wino_f16x3")]:
        MODE.update(m=m, tile=t if t != "-" else "F2", policy="all")
        e = float((conv2d(x, w, None, 1, 1).double() - ref).abs().max())
        rec["single_conv"]["%s/%s" % (m, t)] = e
        print("single conv 192->... %-10s %-4s max err vs fp64 %.2e" % (m, t, e), flush=True)
    for name in a.nets.split(","):
        h = a.size if "8X" not in name else max(8, a.size // 2 // 4 * 4)
        cfg = preset(name)
        for seed in [int(s) for s in a.seeds.split(",")]:
            p = make_params(cfg, seed)
            p64 = {k: v.double() for k, v in p.items()}
            g = torch.Generator().manual_seed(seed + 1)
            lr = torch.rand(2, 3, h, h, generator=g)
            eps = [torch.randn(s, generator=g) * 0.8 for s in eps_shapes(cfg, 2, h, h)]
            inv = O.sr_inverse if cfg.sr else O.rescale_inverse
            res = {}
            with torch.no_grad():
                MODE.update(m="exact")
                ref64 = inv(lr.double(), p64, cfg, 0.8, [e.double() for e in eps], clamp=False)
                runs = [("fp32", "exact", "F2", "all"), ("f16x3 direct", "f16x3", "F2", "all")]
                for t in tiles:
                    for pol in (("all",) if t == "F2" else ("all", "trunk")):
                        runs.append(("%s f32 %s" % (t, pol), "wino_f32", t, pol))
                        runs.append(("%s f16x3 %s" % (t, pol), "wino_f16x3", t, pol))
                for label, m, t, pol in runs:
                    MODE.update(m=m, tile=t, policy=pol)
                    out = inv(lr, p, cfg, 0.8, eps, clamp=False).double()
                    res[label] = float((out - ref64).abs().max())
                    print("%-18s seed %d LR %dx%d scale %.2f  %-22s %.2e" % (name, seed, h, h, float(ref64.abs().max()), label,
                                                                                res[label]), flush=True)
            rec["nets"]["%s/seed%d/lr%d" % (name, seed, h)] = res
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rec, f, indent=1)
