import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The parity tests compare against fixtures at the tolerances of the exact fp32-MFMA kernels unless they select a precision
# themselves; the library default ("f16x3") is pinned by tests/test_cabi_cpu.py::test_default_precision_and_policies.
os.environ.setdefault("HCFLOW_PRECISION", "exact")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
