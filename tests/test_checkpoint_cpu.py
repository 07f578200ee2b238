"""Checkpoint fidelity on the CPU side of the boundary: what BaseModel.save_network / load_network
(codes/models/base_model.py:79-120) do with the drop-in module -- state_dict() -> .cpu() -> torch.save, and
torch.load -> strip a leading 'module.' (checkpoints written from a DataParallel / DDP wrapper) -> load_state_dict(strict)."""
import collections
import os

import pytest
import torch

from hcflow_amd.config import param_spec, preset
from hcflow_amd.params import make_params


def _save_network(network, path):                      # base_model.py:79-95 without the directory housekeeping
    if isinstance(network, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        network = network.module
    state_dict = network.state_dict()
    for key, param in state_dict.items():
        state_dict[key] = param.cpu()
    torch.save(state_dict, path)


def _load_network(path, network, strict=True):        # base_model.py:97-120
    if isinstance(network, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        network = network.module
    load_net = torch.load(path)
    clean = collections.OrderedDict()
    for k, v in load_net.items():
        clean[k[7:] if k.startswith("module.") else k] = v
    network.load_state_dict(clean, strict=strict)


@pytest.mark.parametrize("name", ["SR_4X_tiny", "SR_8X_tiny", "Rescaling_4X_tiny", "SR_DF2K_4X"])
def test_save_load_round_trip_is_strict_and_lossless(name, tmp_path):
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    cfg = preset(name)
    cls = HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling
    src = cls(opt=cfg.to_opt(), step=0)
    src.load_state_dict(make_params(cfg, 7), strict=True)
    path = os.path.join(tmp_path, "latest_G.pth")
    _save_network(src, path)
    saved = torch.load(path)
    want = [(k, tuple(s)) for k, s, _ in param_spec(cfg)]
    assert [(k, tuple(v.shape)) for k, v in saved.items()] == want          # keys, shapes AND order of the reference
    dst = cls(opt=cfg.to_opt(), step=0)
    _load_network(path, dst, strict=True)
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    # a checkpoint written from the wrapper itself carries 'module.' prefixes (the released HCFlow files do not, older
    # BasicSR-style ones do): load_network strips them
    torch.save(collections.OrderedDict(("module." + k, v) for k, v in saved.items()), path)
    dst2 = cls(opt=cfg.to_opt(), step=0)
    _load_network(path, dst2, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst2.state_dict().values()))
    # strictness: a missing or a foreign key is an error, as in the reference
    broken = collections.OrderedDict(saved)
    broken.pop(next(iter(broken)))
    torch.save(broken, path)
    with pytest.raises(RuntimeError):
        _load_network(path, cls(opt=cfg.to_opt(), step=0), strict=True)
    broken = collections.OrderedDict(saved)
    broken["flow.layers.1.not_a_parameter"] = torch.zeros(1)
    torch.save(broken, path)
    with pytest.raises(RuntimeError):
        _load_network(path, cls(opt=cfg.to_opt(), step=0), strict=True)


def test_frozen_haar_weights_survive_the_round_trip(tmp_path):
    """Rescaling checkpoints carry the frozen +-1 Haar filters as parameters (Basic.py:455-468, requires_grad False)."""
    from hcflow_amd import HCFlowNet_Rescaling
    cfg = preset("Rescaling_4X_tiny")
    net = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    haar = [(k, v) for k, v in net.named_parameters() if k.endswith("haar_weights")]
    assert len(haar) == 2 and all(not v.requires_grad for _, v in haar)
    path = os.path.join(tmp_path, "g.pth")
    _save_network(net, path)
    net2 = HCFlowNet_Rescaling(opt=cfg.to_opt(), step=0)
    _load_network(path, net2)
    for (k, a), (_, b) in zip(haar, [(k, v) for k, v in net2.named_parameters() if k.endswith("haar_weights")]):
        assert torch.equal(a, b) and set(a.unique().tolist()) == {-1.0, 1.0}


@pytest.mark.parametrize("name", ["ckpt_sr4_micro", "ckpt_rescale_micro"])
def test_reference_written_checkpoint_loads_strictly_on_cpu(name, tmp_path):
    """The fixture holds the state dict of the REFERENCE module under nn.DataParallel ('module.' keys,
    tests/golden/make_golden.py::gen_checkpoint_fixture): written with torch.save and read back through load_network's
    prefix stripping into our class with strict=True; the CPU oracle reproduces the reference output stored beside it."""
    import collections as C
    import numpy as np
    from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
    from oracle import hcflow_oracle as O
    from tests.util import load_golden, t, maxdiff, seeded_eps
    g = load_golden(name)
    cfg = preset(str(g["preset"]))
    sd = C.OrderedDict((str(k), t(g["t_%d" % i])) for i, k in enumerate(g["keys"]))
    path = os.path.join(tmp_path, "ref_G.pth")
    torch.save(sd, path)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    _load_network(path, torch.nn.DataParallel(net), strict=True)
    want = [(k, tuple(s)) for k, s, _ in param_spec(cfg)]
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == want
    p = {k: v.clone() for k, v in net.state_dict().items()}
    lr = t(g["lr"])
    B, _, h, w = lr.shape
    with torch.no_grad():
        eps = seeded_eps(cfg, B, h, w, 0.8, int(g["eps_seed"]))
        out = (O.sr_inverse if cfg.sr else O.rescale_inverse)(lr, p, cfg, 0.8, eps)
    assert maxdiff(out, g["inv_out"]) <= 1e-5
