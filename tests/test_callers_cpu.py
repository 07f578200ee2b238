"""The reference's own callers ran on top of the drop-in classes in the build container (tests/golden/run_reference_callers.py:
options.parse on the real yml files -> create_model -> define_G -> our class, load_network(strict) of a reference-written
checkpoint, save_network back into the reference, test() / optimize_parameters(step) to the end). This file holds the recorded
call surface to our signatures on the CPU; tests/test_gpu_callers.py replays the recorded calls through the engine."""
import inspect

import numpy as np
import pytest
import torch

from hcflow_amd import HCFlowNet_SR, HCFlowNet_Rescaling
from hcflow_amd.config import param_spec
from tests.util import load_golden, caller_calls, caller_cfg

TEST_TAGS = ["test_sr4", "test_sr8", "test_rescale"]
TRAIN_TAGS = ["train_sr4", "train_rescale"]


@pytest.mark.parametrize("tag", TEST_TAGS + TRAIN_TAGS)
def test_every_recorded_keyword_is_part_of_our_forward_signature(tag):
    g = load_golden("callers_test" if tag.startswith("test") else "callers_train")
    calls = caller_calls(g, tag)
    assert calls
    for cls, training, grad, refusal, kw in calls:
        klass = {"HCFlowNet_SR": HCFlowNet_SR, "HCFlowNet_Rescaling": HCFlowNet_Rescaling}[cls]
        params = inspect.signature(klass.forward).parameters
        assert set(kw) <= set(params), (tag, set(kw) - set(params))
        assert refusal == "HcfError"                       # on a CPU-only host the class refused: no silent fallback
        assert ("hr" in kw) != bool(kw.get("reverse"))     # forward calls carry hr, reverse calls lr only
        if tag.startswith("test"):
            assert not training and not grad and kw.get("training") is False        # netG.eval() + torch.no_grad()
        else:
            assert training and grad


def test_test_driver_call_sequence_matches_the_yml_heats():
    g = load_golden("callers_test")
    for tag in TEST_TAGS:
        calls = caller_calls(g, tag)
        heats = [float(h) for h in g[tag + "_heats"]]
        n = int(g[tag + "_n_sample"])
        assert calls[0][4].get("reverse") is False
        assert [float(c[4]["eps_std"]) for c in calls[1:]] == [h for h in heats for _ in range(n)]
        assert all(c[4]["z"] is None and c[4]["u"] is None for c in calls[1:])


@pytest.mark.parametrize("tag", TRAIN_TAGS)
def test_optimizer_parameter_groups_follow_our_named_parameters(tag):
    """HCFlow_SR_model.py:104-125: every requires_grad parameter of netG in named_parameters order (the frozen Haar filters
    are skipped and reported 'will not optimize')."""
    g = load_golden("callers_train")
    cfg = caller_cfg(g, tag)
    net = (HCFlowNet_SR if cfg.sr else HCFlowNet_Rescaling)(opt=cfg.to_opt(), step=0)
    want = [k for k, p in net.named_parameters() if p.requires_grad]
    assert [str(k) for k in g[tag + "_optim_keys"]] == want
    assert len(want) == len(param_spec(cfg)) - sum(1 for k, _, kind in param_spec(cfg) if kind == "haar")
    lr, b1, b2, wd, eps = [float(v) for v in g[tag + "_optim_hyper"]]
    assert (lr, b1, b2, wd, eps) == (2.5e-4, 0.9, 0.99, 0.0, 1e-8)


@pytest.mark.parametrize("tag,name", [("test_sr4", "SR_DF2K_4X"), ("test_sr8", "SR_CelebA_8X"), ("test_rescale", "Rescaling_DF2K_4X")])
def test_presets_equal_the_reference_yml_network_blocks(tag, name):
    """The three presets (what bench.py, the fixtures and the oracle's eps shapes are built from) against the reference's OWN parse of
    its shipped yml files: run_reference_callers.py recorded NetConfig.from_opt(options.parse(codes/options/test/test_*.yml)) in the
    build container. A typo in a preset -- K, the split positions, RRDB_nb, the widths -- would be shared by product and oracle and
    invisible to every GPU parity test (parity footnote a); here it is not."""
    from hcflow_amd.config import preset, eps_shapes, param_spec
    g = load_golden("callers_test")
    ref = caller_cfg(g, tag)
    ours = preset(name)
    assert repr(ref) == repr(ours)
    assert ref.to_opt() == ours.to_opt()
    assert [k for k, _, _ in param_spec(ref)] == [k for k, _, _ in param_spec(ours)]
    assert eps_shapes(ref, 2, 16, 24) == eps_shapes(ours, 2, 16, 24)
