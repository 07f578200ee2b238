"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hcflow.h declares, and agrees with the Python side (and hence with the reference, see
tests/golden/make_golden.py::build) on the state_dict key/shape table. No compute calls."""
import os
import re

import pytest
import torch

from hcflow_amd import _lib
from hcflow_amd.config import NetConfig, preset, param_spec, param_count, eps_shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRESETS = ["SR_DF2K_4X", "SR_CelebA_8X", "Rescaling_DF2K_4X", "SR_4X_tiny", "SR_8X_tiny", "Rescaling_4X_tiny"]


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "hcflow.h")).read()
    declared = sorted(set(re.findall(r"\b(hcf_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), "libhcflow_hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == declared


@pytest.mark.parametrize("name", PRESETS)
def test_engine_param_table_matches_python(name):
    cfg = preset(name)
    eng = _lib.Engine(cfg)                       # hcf_create touches no device
    want = [(k, tuple(s)) for k, s, _ in param_spec(cfg)]
    assert eng.param_spec() == want


def test_published_param_counts():
    # SR x4 23 232 539, Face x8 27 017 723, Rescale 4 396 350 parameters
    assert param_count(preset("SR_DF2K_4X")) == 23232539
    assert param_count(preset("SR_CelebA_8X")) == 27017723
    assert param_count(preset("Rescaling_DF2K_4X")) == 4396350
    assert len(param_spec(preset("SR_DF2K_4X"))) == 1478


def test_set_param_rejects_bad_keys_and_shapes():
    eng = _lib.Engine(preset("SR_4X_tiny"))
    with pytest.raises(_lib.HcfError):
        eng.set_param("flow.layers.1.nope", torch.zeros(3))
    with pytest.raises(_lib.HcfError):
        eng.set_param("flow.layers.1.actnorm.bias", torch.zeros(1, 13, 1, 1))
    eng.set_param("flow.layers.1.actnorm.bias", torch.zeros(1, 12, 1, 1))


def test_module_surface_and_loud_failure_without_gpu():
    from hcflow_amd.arch import HCFlowNet_SR, HCFlowNet_Rescaling, ActNorm2d
    from hcflow_amd.params import make_params
    for name, cls in (("SR_4X_tiny", HCFlowNet_SR), ("Rescaling_4X_tiny", HCFlowNet_Rescaling)):
        cfg = preset(name)
        net = cls(opt=cfg.to_opt(), step=0)
        sd = net.state_dict()
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s, _ in param_spec(cfg)]
        net.load_state_dict(make_params(cfg, 1), strict=True)
        an = [m for n_, m in net.named_modules() if "ActNorm" in type(m).__name__]
        assert an and all(hasattr(m, "inited") for m in an)
        for m in an:
            m.inited = True                       # HCFlow_SR_model.set_actnorm_init
        net.eval()
        if not torch.cuda.is_available():
            with torch.no_grad(), pytest.raises(_lib.HcfError):
                net(lr=torch.rand(1, 3, 4, 4), eps_std=0.0, reverse=True)
    # frozen Haar weights are not trainable (Basic.py:467-468)
    assert not net.flow.layers[0].haar_weights.requires_grad


def test_from_opt_roundtrip_and_eps_shapes():
    import yaml
    for name in PRESETS:
        cfg = preset(name)
        cfg2 = NetConfig.from_opt(cfg.to_opt())
        assert param_spec(cfg) == param_spec(cfg2)
    assert eps_shapes(preset("SR_DF2K_4X"), 16, 160, 160) == [(16, 21, 160, 160), (16, 6, 320, 320)]
    assert eps_shapes(preset("SR_CelebA_8X"), 2, 20, 20) == [(2, 45, 20, 20), (2, 12, 40, 40), (2, 6, 80, 80)]
