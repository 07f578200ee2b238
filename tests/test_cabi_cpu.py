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


def test_default_precision_and_policies(monkeypatch):
    """The drop-in default is the mode bench.py reports as its headline: f16x3 with the synchronous range check (an
    overflowed pass is re-run exactly); HCFLOW_PRECISION / HCFLOW_RANGE_CHECK override it for unmodified driver scripts."""
    from hcflow_amd.arch import HCFlowNet_SR
    cfg = preset("SR_4X_tiny")
    monkeypatch.delenv("HCFLOW_PRECISION", raising=False)
    monkeypatch.delenv("HCFLOW_RANGE_CHECK", raising=False)
    monkeypatch.delenv("HCFLOW_STREAMS", raising=False)
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    assert net._precision[0] == "f16x3" and net._range_check[0] == "sync"
    # batched inference calls run as two half batches on the process' two side streams by default; B < 4 never splits
    assert net._nstreams[0] == 2 and net._parts(None, 0, None, 3) is None
    assert net.set_streams(1)._parts(None, 0, None, 16) is None
    monkeypatch.setenv("HCFLOW_STREAMS", "1")
    assert HCFlowNet_SR(opt=cfg.to_opt(), step=0)._nstreams[0] == 1
    # the per-call parameter stamp is two flat lists of ints (no per-parameter containers for the garbage collector)
    st = net._stamp_of(net._tensor_list())
    assert len(st) == 2 and all(isinstance(v, int) for v in st[0] + st[1]) and len(st[0]) == len(net._tensors())
    monkeypatch.setenv("HCFLOW_PRECISION", "exact")
    assert HCFlowNet_SR(opt=cfg.to_opt(), step=0)._precision[0] == "exact"
    net.set_precision("exact").set_range_check("lazy")
    assert net._precision[0] == "exact" and net._range_check[0] == "lazy"
    with pytest.raises(AssertionError):
        net.set_range_check("sometimes")


def test_parameters_resolve_through_attributes_on_dataparallel_replicas():
    """nn.DataParallel replicas (torch >= 1.5) have no registered parameters: the module must find its tensors through the
    attribute tree (hcflow_amd/arch.py: _tensors), in state_dict order, and see requires_grad on them."""
    from hcflow_amd.arch import HCFlowNet_SR
    cfg = preset("SR_4X_tiny")
    net = HCFlowNet_SR(opt=cfg.to_opt(), step=0)
    keys = [k for k, _, _ in param_spec(cfg)]
    got = net._tensors()
    assert [k for k, _ in got] == keys
    sd = dict(net.named_parameters())
    assert all(t is sd[k] for k, t in got)
    replica = net._replicate_for_data_parallel()          # what torch.nn.parallel.replicate starts from
    for m in replica.modules():                           # ... and then strips, re-attaching plain tensors
        for name, p_ in list(m._parameters.items()):
            if p_ is not None:
                m._parameters[name] = None
                object.__setattr__(m, name, p_.detach().clone().requires_grad_(p_.requires_grad) * 1.0)
    assert len(list(replica.parameters())) == 0
    rt = replica._tensors()
    assert [k for k, _ in rt] == keys and all(torch.is_tensor(t) for _, t in rt)
    assert replica._wants_grad() and rt[0][1].device.type == "cpu"
