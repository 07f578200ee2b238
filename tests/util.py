"""Shared helpers for the parity tests."""
import os
import functools

import numpy as np
import torch

from hcflow_amd.config import preset
from hcflow_amd.params import make_params, param_digest, digest_close

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@functools.lru_cache(maxsize=4)
def cached_params(preset_name, seed):
    return make_params(preset(preset_name), int(seed))


def params_for(g):
    """Regenerate the fixture's weights from the seeded recipe and check the stored digest."""
    name = str(g["preset"])
    p = cached_params(name, int(g["seed"]))
    if "digest" in g.files:
        d = g["digest"]
        want = {"n": d[0], "sum": d[1], "sumsq": d[2], "probe": d[3]}
        assert digest_close(param_digest(p), want), "seeded parameter recipe drifted from the fixture"
    return preset(name), p


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max())
