import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The suite runs the SHIPPED module default (f16x3) wherever a test does not select a precision itself; the reference-fixture
# modules below are run a second time with HCFLOW_PRECISION=exact (one session-level parametrisation, `hcf_default_precision`),
# so every reference-generated fixture is held against both conv back ends. HCF_TEST_PRECISIONS=f16x3 (or exact) narrows it.
DUAL_PRECISION_MODULES = ("test_gpu_nets", "test_gpu_ops", "test_gpu_backward", "test_gpu_real", "test_gpu_callers",
                          "test_gpu_lu")
_PRECISIONS = [p for p in os.environ.get("HCF_TEST_PRECISIONS", "f16x3,exact").split(",") if p]


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if (mod in DUAL_PRECISION_MODULES and metafunc.definition.get_closest_marker("gpu") is not None
            and "precision" not in metafunc.fixturenames):
        only = metafunc.definition.get_closest_marker("precisions")     # e.g. semantics only the exact fp32-MFMA kernels define
        modes = [p for p in _PRECISIONS if only is None or p in only.args] or list(only.args)
        metafunc.parametrize("hcf_default_precision", modes, indirect=True)


@pytest.fixture(autouse=True)
def hcf_default_precision(request, monkeypatch):
    """The module default the test's nets are constructed with (hcflow_amd/arch.py reads HCFLOW_PRECISION in _setup) and the
    process-wide precision of the per-op entry points."""
    mode = getattr(request, "param", None)
    if mode is None:
        monkeypatch.delenv("HCFLOW_PRECISION", raising=False)
        yield "f16x3"
        return
    monkeypatch.setenv("HCFLOW_PRECISION", mode)
    if request.node.get_closest_marker("gpu") is not None:
        from hcflow_amd import ops
        ops.set_precision(mode)
        yield mode
        ops.set_precision("exact")
    else:
        yield mode


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "precisions(*modes): restrict the session-level conv-precision parametrisation of this test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
