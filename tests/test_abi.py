"""CPU: the C-ABI library loads and exports every symbol include/qbm_hip.h declares, with the signatures
the ctypes binding uses (no compute call: there is no GPU here); argument validation returns error codes."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from qbm_image_classification_amd import _lib
    from qbm_image_classification_amd.build import build_library

    build_library()          # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "qbm_hip.h")).read()
    return sorted(set(re.findall(r"QBM_API\s+[\w\s\*]+?\b(qbm_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from qbm_image_classification_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in qbm_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.qbm_abi_version() == _lib.QBM_ABI_VERSION == 15


def test_argument_validation_without_a_gpu(lib):
    from qbm_image_classification_amd._lib import RbmParams

    bad = RbmParams(None, None, None, None, None, 16, 8, 2)
    assert lib.qbm_rbm_sample_hidden(C.byref(bad), None, None, 4, None, None, 0, 0, 0, None) == -1
    assert b"null parameter block" in lib.qbm_last_error()
    assert lib.qbm_qubo_gibbs(None, None, 2, 2, 5000, 1, 1.0, 1.0, 0, 0, None, None, None) == -1
    assert b"n out of range" in lib.qbm_last_error()
    assert lib.qbm_qubo_gibbs(None, None, 0, 5, 8, 1, 1.0, 1.0, 0, 0, None, None, None) == 0   # empty batch is a no-op
    assert lib.qbm_f32_to_u8(None, None, 0, None) == 0


def test_product_has_no_cpu_fallback():
    import torch

    from qbm_image_classification_amd import QbmHipError
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(QbmHipError):
        ClassificationRBM(16, 8, 1)
    import glob

    mods = glob.glob(os.path.join(ROOT, "qbm_image_classification_amd", "*.py")) + [os.path.join(ROOT, "rbm_main.py"),
                                                                                    os.path.join(ROOT, "qbm_main.py")]
    assert len(mods) >= 12
    for mod in mods:     # nothing shipped may import, call or link the oracle
        text = open(mod).read()
        assert "import oracle" not in text and "from oracle" not in text and "qbm_oracle" not in text, mod
    # the HIP-backed entry points of the "next" rows refuse to run without the library / device too
    from qbm_image_classification_amd import coreset, device_data

    with pytest.raises(QbmHipError):
        device_data.HipDataOps("cuda:0")
    with pytest.raises(QbmHipError):
        coreset.k_center_greedy(__import__("numpy").zeros((8, 2), "float32"), 3, random_seed=0)
