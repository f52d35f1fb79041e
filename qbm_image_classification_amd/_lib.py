"""ctypes binding of libqbm_hip.so (include/qbm_hip.h).  There is no CPU fallback: if the HIP
library is missing or a call fails, this module raises."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# QBM_HIP_LIB: another build of the SAME library (a diagnostic / experiment build of csrc/, e.g. tools/stamp_kernels.py's): a
# developer's switch for A/B timing inside one GPU session, never a fallback -- the file must exist and pass the ABI check
SO = os.environ.get("QBM_HIP_LIB") or os.path.join(HERE, "libqbm_hip.so")

QBM_ABI_VERSION = 15
QBM_MAX_CLASSES = 15
QBM_F32, QBM_U8 = 0, 1


class QbmHipError(RuntimeError):
    pass


class RbmParams(C.Structure):
    _fields_ = [("W", C.c_void_p), ("b", C.c_void_p), ("d", C.c_void_p), ("U", C.c_void_p), ("c", C.c_void_p),
                ("V", C.c_int), ("H", C.c_int), ("C", C.c_int)]


class RbmGrads(C.Structure):
    _fields_ = [("gW", C.c_void_p), ("gb", C.c_void_p), ("gd", C.c_void_p), ("gU", C.c_void_p), ("gc", C.c_void_p),
                ("n_rows", C.c_void_p)]   # optional: receives the row count of the gradient launch (data-parallel path)


class CdWorkspace(C.Structure):
    _fields_ = [("ph0", C.c_void_p), ("h", C.c_void_p), ("vk", C.c_void_p), ("yk", C.c_void_p), ("phk", C.c_void_p),
                ("sync", C.c_void_p),   # state of the fused chain kernel (QBM_CD_SYNC_WORDS uint32, zeroed once) or None
                # the training loop's monitor (include/qbm_hip.h): labels, x.W + d, p(y|x), per-row loss terms, running sums
                ("mon_labels", C.c_void_p), ("mon_pre", C.c_void_p), ("mon_p", C.c_void_p), ("mon_rowloss", C.c_void_p),
                ("mon_loss_acc", C.c_void_p),
                ("draw_base", C.c_void_p)]   # optional [2] uint64 device words added to (offset, row0) by every sampling launch


QBM_CD_SYNC_WORDS = 3136


class DiscWorkspace(C.Structure):
    _fields_ = [("pre", C.c_void_p), ("diff", C.c_void_p), ("rowloss", C.c_void_p), ("loss_acc", C.c_void_p),
                ("status", C.c_void_p)]   # optional: status word of a CD workspace (hybrid training's fail-safe)


_p = C.c_void_p
_i = C.c_int
_u64 = C.c_uint64
_u32 = C.c_uint32
_f = C.c_float
_PP = C.POINTER(RbmParams)
_GP = C.POINTER(RbmGrads)

# symbol -> argtypes; every symbol declared in include/qbm_hip.h must appear here (tests check it)
SIGNATURES = {
    "qbm_abi_version": [],
    "qbm_last_error": [],
    "qbm_rbm_sample_hidden": [_PP, _p, _p, _i, _p, _p, _u64, _u64, _u32, _p],
    "qbm_rbm_sample_hidden_ex": [_PP, _p, _i, _p, _i, _p, _p, _p, _u64, _u64, _u32, _p],
    "qbm_rbm_hidden_preact": [_PP, _p, _i, _p, _p],
    "qbm_rbm_sample_visible": [_PP, _p, _i, _p, _p, _u64, _u64, _u32, _p],
    "qbm_rbm_sample_class": [_PP, _p, _i, _p, _p, _u64, _u64, _u32, _p],
    "qbm_rbm_free_energy": [_PP, _p, _i, _p, _p, _p, _p],
    "qbm_rbm_two_phase_grad": [_i, _i, _i, _i, _p, _i, _p, _p, _p, _i, _p, _p, _GP, _p, _PP, _f, _f, _f, _f, _p],
    "qbm_rbm_sample_visible_class_ex": [_PP, _p, _i, _i, _p, _p, _p, _p, _p, _u64, _u64, _u64, _u32, _p],
    "qbm_f32_to_u8": [_p, _p, C.c_long, _p],
    "qbm_rbm_update": [_PP, _GP, _f, _f, _f, _f, _p],
    "qbm_rbm_update_dev": [_PP, _GP, _f, _f, _f, _p, _f, _p, _p],
    "qbm_rbm_fused_chain_mode": [_i],
    "qbm_rbm_cd_chain": [_PP, _p, _i, _p, _i, _i, _u64, _u64, _u32, _p, _p, C.POINTER(CdWorkspace), _p],
    "qbm_rbm_cd_step": [_PP, _GP, _p, _i, _p, _i, _i, _u64, _u64, _u32, _p, _p, C.POINTER(CdWorkspace), _i, _f, _f,
                        _f, _f, _p],
    "qbm_rbm_dp_step": [_PP, _GP, _p, _i, _p, _i, _i, _u64, _u64, _u32, _p, _p, C.POINTER(CdWorkspace), _i, _f, _f, _f, _p],
    "qbm_rbm_cd_step_finish": [_PP, _GP, _p, _i, _p, _i, _i, _u64, _u64, _u32, _p, _p, C.POINTER(CdWorkspace), _i, _f, _f,
                               _f, _f, _u32, _p],
    "qbm_rbm_cd_tail": [_PP, _GP, _p, _i, _p, _i, _p, _p, C.POINTER(CdWorkspace), _i, _f, _f, _f, _f, _u32, _p],
    "qbm_test_hgrad_fault": [_i, _i],
    "qbm_rbm_disc_step": [_PP, _GP, _p, _p, _i, _p, _p, C.POINTER(DiscWorkspace), _i, _i, _f, _f, _f, _f, _p],
    "qbm_rbm_disc_step_ex": [_PP, _GP, _p, _i, _p, _i, _p, _p, C.POINTER(DiscWorkspace), _i, _i, _f, _f, _f, _f, _p],
    "qbm_data_binarise": [_p, C.c_long, _f, _p, _p],
    "qbm_data_pack_bits": [_p, _i, _i, _p, _p],
    "qbm_data_shuffle_keys": [_u64, _u64, _i, _p, _p],
    "qbm_data_gather": [_p, _i, _p, _i, _i, _p, _i, _p, _p, _p, _i, _p],
    "qbm_kcenter_workspace_bytes": [_i, _i],
    "qbm_kcenter_greedy": [_p, _i, _i, C.POINTER(C.c_int32), _i, _i, _p, _p, _p, _p],
    "qbm_kcenter_fused_workspace_bytes": [],
    "qbm_kcenter_greedy_fused": [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p],
    "qbm_test_math": [_i, _p, _p, _i, _p],
    "qbm_test_uniform": [_u64, _u64, _u32, _i, _i, _p, _p],
    "qbm_qubo_gibbs": [_p, _p, _i, _i, _i, _i, _f, _f, _u64, _u64, _p, _p, _p],
    "qbm_qubo_gibbs_ex": [_p, _p, _i, _i, _i, _i, _f, _f, _u64, _u64, _u32, _p, _p, _p],
    "qbm_qubo_gibbs_dev": [_p, _p, _i, _i, _i, _i, _p, _f, _f, _u64, _u64, _u32, _p, _p, _p],
    "qbm_qubo_from_Q": [_p, _i, _i, _p, _p, _p, _p, _p, _p],
    "qbm_u8_to_f32": [_p, _p, C.c_long, _p],
    "qbm_qubo_csr_max_entries": [_i, _i],
    "qbm_qubo_csr_workspace_bytes": [_i, C.c_long],
    "qbm_qubo_csr_build": [_p, _i, C.c_long, _p, _p],
    "qbm_qubo_gibbs_csr": [_p, _p, C.c_long, _i, _i, _i, _i, _p, _f, _f, _u64, _u64, _u32, _p, _p, _p],
    "qbm_qbm_linear_fields_dev": [_p, _p, _p, _i, _i, _i, C.c_double, _p, _p],
    "qbm_qbm_couplings": [_p, _p, _i, _i, C.c_double, _p, _p, _p, _p],
    "qbm_qbm_beta_range": [_p, _p, _p, _i, _i, _p, _p],
    "qbm_qbm_stats_diff": [_p, _p, _p, _p, _i, _i, _i, _p, _p],
    "qbm_qbm_stats_diff_nll": [_p, _p, _p, _p, _i, _i, _i, _p, _p],
    "qbm_qbm_apply_update": [_p, _p, _i, _i, C.c_double, C.c_double, _p],
    "qbm_qbm_batch_stats": [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p],
    "qbm_qbm_linear_fields": [_p, _p, _p, _p, _p, C.c_double, _i, _i, _i, C.c_double, _p, _p],
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise loudly if it has not been built (python -m ...build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise QbmHipError(
            f"{SO} not found: build the HIP extension first (python -m qbm_image_classification_amd.build). "
            "There is no CPU fallback for this path.")
    # torch must map its HIP runtime first: libqbm_hip.so then binds to that same libamdhip64 instance
    # (streams and device pointers are shared with torch); loading ours first gives two runtimes.
    import torch  # noqa: F401

    lib = C.CDLL(SO)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "qbm_last_error" else C.c_long if name.endswith(("_bytes", "_max_entries")) else C.c_int
    if lib.qbm_abi_version() != QBM_ABI_VERSION:
        raise QbmHipError(f"ABI mismatch: library {lib.qbm_abi_version()} vs binding {QBM_ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().qbm_last_error()
        raise QbmHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  The tensor must be a contiguous CUDA tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise QbmHipError("expected a ROCm device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise QbmHipError("expected a contiguous tensor")
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
