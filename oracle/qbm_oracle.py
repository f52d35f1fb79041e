"""oracle/qbm_oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes/numpy front end of oracle/qbm_oracle.c, the CPU restatement of the reference's
Gibbs / contrastive-divergence hot path (see the header of qbm_oracle.c for the numerical
contract and the reference file:line each function follows).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (qbm_image_classification_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqbm_oracle.so")

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, seconds). Building the checker is not using it."""
    src = os.path.join(_HERE, "qbm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        with open("/proc/cpuinfo") as f:
            flags = f.read()
        if " fma" not in flags or " avx2" not in flags:
            raise RuntimeError("oracle needs a CPU with AVX2+FMA (exact fmaf chains)")
        _lib = C.CDLL(_SO)
    return _lib


def _opt(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def philox_uniform(seed: int, offset: int, row0: int, rows: int, cols: int) -> np.ndarray:
    out = np.empty((rows, cols), np.float32)
    f = lib().orc_philox_uniform
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, _f32p]
    f.restype = None
    f(seed, offset, row0, rows, cols, out)
    return out


def philox_raw(key, ctr) -> np.ndarray:
    k = np.asarray(key, dtype=np.uint32)
    c = np.asarray(ctr, dtype=np.uint32)
    out = np.empty(4, np.uint32)
    u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
    f = lib().orc_philox_raw
    f.argtypes = [u32p, u32p, u32p]
    f.restype = None
    f(k, c, out)
    return out


def data_shuffle_keys(seed: int, epoch: int, n: int) -> np.ndarray:
    out = np.empty(n, np.int32)
    f = lib().orc_shuffle_keys
    f.argtypes = [C.c_uint64, C.c_uint64, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")]
    f.restype = None
    f(seed, epoch, n, out)
    return out


def data_shuffle_perm(seed: int, epoch: int, n: int) -> np.ndarray:
    """Epoch permutation of the device-resident dataset: stable ascending argsort of the Philox keys."""
    return np.argsort(data_shuffle_keys(seed, epoch, n), kind="stable").astype(np.int64)


def data_gather(x, idx, labels=None, num_classes=0, bits_visible=None, out_dtype=np.float32):
    """Minibatch gather (+ unpacking of MSB-first bit-packed rows when bits_visible is given) with labels / one-hot."""
    x = np.asarray(x)
    rows = x[np.asarray(idx)]
    if bits_visible is not None:
        rows = np.unpackbits(rows.astype(np.uint8), axis=1)[:, :bits_visible]
    rows = rows.astype(out_dtype)
    if labels is None:
        return rows
    lab = np.asarray(labels)[np.asarray(idx)].astype(np.int32)
    onehot = (lab[:, None] == np.arange(num_classes)[None, :]).astype(np.float32)
    return rows, lab, onehot


def kcenter_picks(emb, seeds, n_picks):
    """orc_kcenter_greedy: (picks int32 [n_picks], selected uint8 [N]) for given seed centres."""
    emb = _f32(emb)
    n, d = emb.shape
    seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    picks = np.empty(n_picks, np.int32)
    sel = np.empty(n, np.uint8)
    i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    f = lib().orc_kcenter_greedy
    f.argtypes = [_f32p, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p, np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")]
    f.restype = None
    f(emb, n, d, seeds, len(seeds), n_picks, picks, sel)
    return picks, sel


def k_center_greedy(embedding_matrix, budget, random_seed=None, index=None, already_selected=None):
    """src/coreset.py:28-106 restated around orc_kcenter_greedy (budget bookkeeping :38-44, seed centre :64-67,
    result = index[select_mask] :106)."""
    emb = _f32(embedding_matrix)
    n = emb.shape[0]
    budget = min(budget + 1, n)
    index = np.arange(n) if index is None else np.asarray(index)
    sel = [] if already_selected is None else list(already_selected)
    np.random.seed(random_seed)
    if len(sel) == 0:
        sel = [int(np.random.randint(0, n))]
    mask = np.zeros(n, bool)
    mask[sel] = True
    remaining = budget - int(mask.sum())
    if remaining <= 0:
        return index[mask]
    _, selected = kcenter_picks(emb, np.flatnonzero(mask), remaining)
    return index[selected.astype(bool)]


def math(which: str, x) -> np.ndarray:
    x = _f32(x)
    y = np.empty_like(x)
    f = lib().orc_math
    f.argtypes = [C.c_int, _f32p, _f32p, C.c_int]
    f.restype = None
    f({"exp": 0, "log": 1, "sigmoid": 2, "softplus": 3}[which], x.ravel(), y.reshape(-1), x.size)
    return y


def qgemm(A, B, nq=4, perm=None) -> np.ndarray:
    """[Q]: the contraction cut into nq chains (8: conditionals, 4: gradient blocks); perm = piece order [Q'] (default: the
    conditionals' order with 8 chains, ascending with 4)"""
    perm = (nq == 8) if perm is None else perm
    A, B = _f32(A), _f32(B)
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    out = np.empty((M, N), np.float32)
    f = lib().orc_qgemm
    f.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    f.restype = None
    f(A, B, out, M, N, K, int(nq), int(bool(perm)))
    return out


def rbm_sample_hidden(v, y, W, d, U, seed=0, offset=0, row0=0, sample=True):
    v, W, d, U = _f32(v), _f32(W), _f32(d), _f32(U)
    B, V = v.shape
    H = W.shape[1]
    Cn = U.shape[0]
    y = None if y is None else _f32(y)
    p = np.empty((B, H), np.float32)
    s = np.empty((B, H), np.float32) if sample else None
    f = lib().orc_rbm_sample_hidden
    f.argtypes = [_f32p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64,
                  C.c_uint64, C.c_uint32, _f32p, C.c_void_p]
    f.restype = None
    f(v, _opt(y), W, d, U, B, V, H, Cn, seed, offset, row0, p, _opt(s))
    return p, s


def rbm_sample_visible(h, W, b, seed=0, offset=0, row0=0, sample=True):
    h, W, b = _f32(h), _f32(W), _f32(b)
    B, H = h.shape
    V = W.shape[0]
    p = np.empty((B, V), np.float32)
    s = np.empty((B, V), np.float32) if sample else None
    f = lib().orc_rbm_sample_visible
    f.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _f32p,
                  C.c_void_p]
    f.restype = None
    f(h, W, b, B, V, H, seed, offset, row0, p, _opt(s))
    return p, s


def rbm_sample_class(h, U, c, seed=0, offset=0, row0=0, sample=True):
    h, U, c = _f32(h), _f32(U), _f32(c)
    B, H = h.shape
    Cn = U.shape[0]
    p = np.empty((B, Cn), np.float32)
    y = np.empty((B, Cn), np.float32) if sample else None
    f = lib().orc_rbm_sample_class
    f.argtypes = [_f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _f32p,
                  C.c_void_p]
    f.restype = None
    f(h, U, c, B, H, Cn, seed, offset, row0, p, _opt(y))
    return p, y


def rbm_free_energy(x, W, d, U, c):
    """-> (pre [B,H], negF [B,C], p(y|x) [B,C])"""
    x, W, d, U, c = _f32(x), _f32(W), _f32(d), _f32(U), _f32(c)
    B, V = x.shape
    H = W.shape[1]
    Cn = U.shape[0]
    pre = np.empty((B, H), np.float32)
    nf = np.empty((B, Cn), np.float32)
    p = np.empty((B, Cn), np.float32)
    f = lib().orc_rbm_free_energy
    f.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p]
    f.restype = None
    f(x, W, d, U, c, B, V, H, Cn, pre, nf, p)
    return pre, nf, p


def rbm_update(W, b, d, U, c, gW, gb, gd, gU, gc, lr, factor, batch_size, sparse):
    """In place on W, b, d, U, c (float32 C-contiguous arrays). gb may be None."""
    V, H = W.shape
    Cn = U.shape[0]
    f = lib().orc_rbm_update
    f.argtypes = [_f32p] * 5 + [_f32p, C.c_void_p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                C.c_float, C.c_float, C.c_float]
    f.restype = None
    gb = None if gb is None else _f32(gb)
    f(W, b, d, U, c, _f32(gW), _opt(gb), _f32(gd), _f32(gU), _f32(gc), V, H, Cn, lr, factor, batch_size, sparse)


def rbm_two_phase_grad(A0, Y0, P0, A1, Y1, P1):
    A0, Y0, P0, A1, Y1, P1 = map(_f32, (A0, Y0, P0, A1, Y1, P1))
    B, V = A0.shape
    H = P0.shape[1]
    Cn = Y0.shape[1]
    gW = np.empty((V, H), np.float32)
    gb = np.empty(V, np.float32)
    gd = np.empty(H, np.float32)
    gU = np.empty((Cn, H), np.float32)
    gc = np.empty(Cn, np.float32)
    f = lib().orc_rbm_two_phase_grad
    f.argtypes = [_f32p] * 6 + [C.c_int] * 4 + [_f32p] * 5
    f.restype = None
    f(A0, Y0, P0, A1, Y1, P1, B, V, H, Cn, gW, gb, gd, gU, gc)
    return gW, gb, gd, gU, gc


def rbm_disc_grad(x, labels, W, d, U, c, with_loss=False):
    """-> (gW, gd, gU, gc, p(y|x), negF); gb is identically zero in the reference (:138).
    with_loss: additionally (rowloss [B,2], batchloss [2]): CrossEntropyLoss-on-probabilities and -log(p_y + 1e-8) per row
    and as batch means (src/ClassificationRBM.py:142, :188-191)."""
    x, W, d, U, c = _f32(x), _f32(W), _f32(d), _f32(U), _f32(c)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    B, V = x.shape
    H = W.shape[1]
    Cn = U.shape[0]
    gW = np.empty((V, H), np.float32)
    gd = np.empty(H, np.float32)
    gU = np.empty((Cn, H), np.float32)
    gc = np.empty(Cn, np.float32)
    p = np.empty((B, Cn), np.float32)
    nf = np.empty((B, Cn), np.float32)
    rl = np.empty((B, 2), np.float32)
    bl = np.empty(2, np.float32)
    f = lib().orc_rbm_disc_grad
    f.argtypes = [_f32p, _i32p, _f32p, _f32p, _f32p, _f32p] + [C.c_int] * 4 + [_f32p] * 8
    f.restype = None
    f(x, labels, W, d, U, c, B, V, H, Cn, gW, gd, gU, gc, p, nf, rl, bl)
    if with_loss:
        return gW, gd, gU, gc, p, nf, rl, bl
    return gW, gd, gU, gc, p, nf


def rbm_class_loss(p, labels):
    """-> (rowloss [B,2], batchloss [2]): per row CrossEntropyLoss ON THE PROBABILITIES (src/ClassificationRBM.py:142) and
    -log(p_y + 1e-8) (:188-190), and their batch means (quartered chain over the rows, / B)."""
    p = _f32(p)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    B, Cn = p.shape
    rl = np.empty((B, 2), np.float32)
    bl = np.empty(2, np.float32)
    f = lib().orc_rbm_class_loss
    f.argtypes = [_f32p, _i32p, C.c_int, C.c_int, _f32p, _f32p]
    f.restype = None
    f(p, labels, B, Cn, rl, bl)
    return rl, bl


def rbm_cd_chain(v0, y0, W, b, d, U, c, k, seed=0, offset=0, row0=0, chain_v=None, chain_y=None):
    """-> (ph0, vk, yk, phk); chain_v / chain_y (PCD state) are updated in place when given."""
    v0, y0, W, b, d, U, c = map(_f32, (v0, y0, W, b, d, U, c))
    B, V = v0.shape
    H = W.shape[1]
    Cn = U.shape[0]
    ph0 = np.empty((B, H), np.float32)
    vk = np.empty((B, V), np.float32)
    yk = np.empty((B, Cn), np.float32)
    phk = np.empty((B, H), np.float32)
    f = lib().orc_rbm_cd_chain
    f.argtypes = [_f32p] * 7 + [C.c_int] * 5 + [C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p] + [
        _f32p] * 4
    f.restype = None
    if chain_v is not None:
        assert chain_v.dtype == np.float32 and chain_v.flags.c_contiguous
        assert chain_y.dtype == np.float32 and chain_y.flags.c_contiguous
    f(v0, y0, W, b, d, U, c, B, V, H, Cn, k, seed, offset, row0, _opt(chain_v), _opt(chain_y), ph0, vk, yk, phk)
    return ph0, vk, yk, phk


def qbm_linear_fields(X, labels, W_vh, W_vo, b_h, b_o, beta_eff=1.0):
    X = np.ascontiguousarray(X, dtype=np.float64)
    W_vh = np.ascontiguousarray(W_vh, dtype=np.float64)
    W_vo = np.ascontiguousarray(W_vo, dtype=np.float64).reshape(-1)
    b_h = np.ascontiguousarray(b_h, dtype=np.float64)
    B, D = X.shape
    H = W_vh.shape[1]
    n = H if labels is not None else H + 1
    lin = np.empty((B, n), np.float32)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
    f = lib().orc_qbm_linear_fields
    f.argtypes = [_f64p, C.c_void_p, _f64p, _f64p, _f64p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, _f32p]
    f.restype = None
    f(X, _opt(lab), W_vh, W_vo, b_h, float(np.asarray(b_o).reshape(-1)[0]), B, D, H, beta_eff, lin)
    return lin


def qbm_couplings(W_vh, W_hh, clamped: bool, beta_eff=1.0):
    W_vh = np.ascontiguousarray(W_vh, dtype=np.float64)
    H = W_vh.shape[1]
    n = H if clamped else H + 1
    J = np.empty((n, n), np.float32)
    whh = None if W_hh is None else np.ascontiguousarray(W_hh, dtype=np.float64)
    f = lib().orc_qbm_couplings
    f.argtypes = [_f64p, C.c_void_p, C.c_int, C.c_int, C.c_double, _f32p]
    f.restype = None
    f(W_vh, _opt(whh), H, int(clamped), beta_eff, J)
    return J


def qbm_coupling_stats(J):
    """rowabs[n] = sum_b |J_ab| (float64, ascending b: cumsum is sequential), rowmin[n] = smallest non-zero |J_ab| or 0:
    the second and third output of qbm_qbm_couplings (include/qbm_hip.h)."""
    A = np.abs(np.asarray(J, dtype=np.float32)).astype(np.float64)
    rowabs = np.cumsum(A, axis=1)[:, -1] if A.shape[1] else np.zeros(A.shape[0])
    masked = np.where(A > 0, A, np.inf)
    rowmin = masked.min(axis=1)
    return rowabs, np.where(np.isfinite(rowmin), rowmin, 0.0)


def qbm_beta_range(lin, rowabs, rowmin):
    """Per-image annealing range (hot, cold) as qbm_qbm_beta_range states it -> float32 [B,2].  max / min are exact in any
    order; the two divisions and the clamp are single float64 operations."""
    lin = np.abs(np.asarray(lin, dtype=np.float32)).astype(np.float64)
    B = lin.shape[0]
    out = np.empty((B, 2), np.float32)
    for b in range(B):
        big = float(np.max(lin[b] + rowabs)) if lin.shape[1] else 0.0
        cands = np.concatenate([lin[b][lin[b] > 0], np.asarray(rowmin)[np.asarray(rowmin) > 0]])
        hot, cold = 0.1, 1.0
        if cands.size and big > 0 and np.isfinite(big):
            mn = float(cands.min())
            hot = 0.6931471805599453 / big
            cold = 4.605170185988092 / mn
            cold = min(max(cold, 10.0 * hot), 1e4 * hot)
        out[b] = (np.float32(hot), np.float32(cold))
    return out


def qbm_stats_diff(X, labels, mean_c, mean_u, H):
    """statistics(clamped) - statistics(unclamped) in the theta layout [W_vh | W_vo | b_h | b_o] (qbm_qbm_stats_diff)."""
    c = qbm_batch_stats(X, labels, mean_c, H)
    u = qbm_batch_stats(X, None, mean_u, H)
    return np.concatenate([a.reshape(-1) for a in c]) - np.concatenate([a.reshape(-1) for a in u])


def qubo_gibbs(lin, J, reads, sweeps, beta0=1.0, beta1=1.0, seed=0, offset=0, chain0=0):
    """-> (states u8 [B,R,n], mean f64 [B,n]); chain0 = global index of the first chain (sharded minibatches)"""
    lin, J = _f32(lin), _f32(J)
    B, n = lin.shape
    states = np.empty((B, reads, n), np.uint8)
    mean = np.empty((B, n), np.float64)
    f = lib().orc_qubo_gibbs
    f.argtypes = [_f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_uint64,
                  C.c_uint32, _u8p, _f64p]
    f.restype = None
    f(lin, J, B, reads, n, sweeps, beta0, beta1, seed, offset, chain0, states, mean)
    return states, mean


def qbm_batch_stats(X, labels, mean, H):
    """get_average_configuration_batch from per-image means -> (W_vh [(1+D),H], W_vo [D], b_h [H], b_o [1]) f64"""
    X = np.ascontiguousarray(X, dtype=np.float64)
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    B, D = X.shape
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32)
    W_vh = np.empty((1 + D, H), np.float64)
    W_vo = np.empty(D, np.float64)
    b_h = np.empty(H, np.float64)
    b_o = np.empty(1, np.float64)
    f = lib().orc_qbm_batch_stats
    f.argtypes = [_f64p, C.c_void_p, _f64p, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p, _f64p]
    f.restype = None
    f(X, _opt(lab), mean, B, D, H, W_vh, W_vo, b_h, b_o)
    return W_vh, W_vo, b_h, b_o
