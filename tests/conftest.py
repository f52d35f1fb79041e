import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement under oracle/ -- the checker, never the thing measured."""
    from oracle import qbm_oracle

    qbm_oracle.build()
    qbm_oracle.lib()
    return qbm_oracle


@pytest.fixture(scope="session", autouse=True)
def _hip_library_present():
    """The built libqbm_hip.so normally travels with the tree; if it does not (fresh checkout), build it -- building
    the product is not a fallback: without hipcc the GPU tests then fail loudly in _lib.load()."""
    from qbm_image_classification_amd import build as b

    if not os.path.exists(b.SO) and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        b.build_library(force=False, verbose=False)
    yield
