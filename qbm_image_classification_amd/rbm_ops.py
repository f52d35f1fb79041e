"""Tensor-level front end of the RBM entry points of libqbm_hip.so (include/qbm_hip.h).

Everything here is plumbing: torch ROCm tensors in, tensors out, device pointers handed to the C ABI
on torch's current stream.  No arithmetic of the hot path happens in Python, and there is no CPU
fallback -- a missing library or a CPU tensor raises QbmHipError."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import QBM_F32, QBM_U8, CdWorkspace, DiscWorkspace, RbmGrads, RbmParams, check, current_stream, ptr


class ParamSet:
    """The five parameter blocks of a ClassificationRBM (src/ClassificationRBM.py:26-30) on the device."""

    def __init__(self, W, b, d, U, c):
        self.W, self.b, self.d, self.U, self.c = W, b, d, U, c
        self.V, self.H = W.shape
        self.C = U.shape[0]
        for t in (W, b, d, U, c):
            if t.dtype != torch.float32:
                raise _lib.QbmHipError("parameters must be float32")

    def struct(self) -> RbmParams:
        return RbmParams(ptr(self.W), ptr(self.b), ptr(self.d), ptr(self.U), ptr(self.c), self.V, self.H, self.C)


class GradSet:
    """One flat fp32 buffer [gW | gb | gd | gU | gc | n_rows] with views: the data-parallel path
    all-reduces `flat` once per step; the last element carries the local row count."""

    def __init__(self, V, H, C, device):
        self.V, self.H, self.C = V, H, C
        n = V * H + V + H + C * H + C
        # 16-byte aligned blocks are not required by the ABI (it falls back to scalar accesses), the
        # flat layout simply follows the reference's attribute order
        self.flat = torch.zeros(n + 1, dtype=torch.float32, device=device)
        o = 0
        self.gW = self.flat[o:o + V * H].view(V, H); o += V * H
        self.gb = self.flat[o:o + V]; o += V
        self.gd = self.flat[o:o + H]; o += H
        self.gU = self.flat[o:o + C * H].view(C, H); o += C * H
        self.gc = self.flat[o:o + C]; o += C
        self.count = self.flat[o:o + 1]

    def struct(self, with_gW=True, with_gb=True) -> RbmGrads:
        return RbmGrads(ptr(self.gW) if with_gW else None, ptr(self.gb) if with_gb else None, ptr(self.gd),
                        ptr(self.gU), ptr(self.gc))


def _f32(t, device):
    t = torch.as_tensor(t)
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


def _state(t, device):
    """A Gibbs-state tensor for the *_ex entry points: uint8 stays uint8, everything else becomes fp32."""
    t = torch.as_tensor(t)
    if t.dtype == torch.uint8:
        if t.device != device or not t.is_contiguous():
            t = t.to(device).contiguous()
        return t, QBM_U8
    return _f32(t, device), QBM_F32


class HipRbmOps:
    """The HIP backend.  (tests/ substitute an oracle-backed double with the same methods to exercise
    the host logic on CPU; the product never does.)"""

    name = "hip"

    def __init__(self, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.QbmHipError("no ROCm device visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device(device)
        self._disc_ws = None

    # ---- conditionals -------------------------------------------------------------------------
    def sample_hidden(self, P: ParamSet, v, y=None, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        v = _f32(v, self.device)
        y = None if y is None else _f32(y, self.device)
        B = v.shape[0]
        p = torch.empty(B, P.H, dtype=torch.float32, device=self.device) if want_p else None
        s = torch.empty(B, P.H, dtype=torch.float32, device=self.device) if want_s else None
        ps = P.struct()
        check(self.lib.qbm_rbm_sample_hidden(C.byref(ps), ptr(v), ptr(y), B, ptr(p), ptr(s), seed, offset, row0,
                                             current_stream()), "qbm_rbm_sample_hidden")
        return p, s

    def hidden_preact(self, P: ParamSet, x):
        x = _f32(x, self.device)
        pre = torch.empty(x.shape[0], P.H, dtype=torch.float32, device=self.device)
        ps = P.struct()
        check(self.lib.qbm_rbm_hidden_preact(C.byref(ps), ptr(x), x.shape[0], ptr(pre), current_stream()),
              "qbm_rbm_hidden_preact")
        return pre

    def sample_visible(self, P: ParamSet, h, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        h = _f32(h, self.device)
        B = h.shape[0]
        p = torch.empty(B, P.V, dtype=torch.float32, device=self.device) if want_p else None
        s = torch.empty(B, P.V, dtype=torch.float32, device=self.device) if want_s else None
        ps = P.struct()
        check(self.lib.qbm_rbm_sample_visible(C.byref(ps), ptr(h), B, ptr(p), ptr(s), seed, offset, row0,
                                              current_stream()), "qbm_rbm_sample_visible")
        return p, s

    def sample_class(self, P: ParamSet, h, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        h = _f32(h, self.device)
        B = h.shape[0]
        p = torch.empty(B, P.C, dtype=torch.float32, device=self.device) if want_p else None
        s = torch.empty(B, P.C, dtype=torch.float32, device=self.device) if want_s else None
        ps = P.struct()
        check(self.lib.qbm_rbm_sample_class(C.byref(ps), ptr(h), B, ptr(p), ptr(s), seed, offset, row0,
                                            current_stream()), "qbm_rbm_sample_class")
        return p, s

    def sample_hidden_ex(self, P: ParamSet, v, y=None, seed=0, offset=0, row0=0, want_p=True, want_s=False,
                         want_s8=True):
        """v may be uint8 {0,1}; the sample can come back as fp32 and/or uint8. -> (p, s_f32, s_u8)"""
        v, vd = _state(v, self.device)
        y = None if y is None else _f32(y, self.device)
        B = v.shape[0]
        p = torch.empty(B, P.H, dtype=torch.float32, device=self.device) if want_p else None
        s = torch.empty(B, P.H, dtype=torch.float32, device=self.device) if want_s else None
        s8 = torch.empty(B, P.H, dtype=torch.uint8, device=self.device) if want_s8 else None
        ps = P.struct()
        check(self.lib.qbm_rbm_sample_hidden_ex(C.byref(ps), ptr(v), vd, ptr(y), B, ptr(p), ptr(s), ptr(s8), seed,
                                                offset, row0, current_stream()), "qbm_rbm_sample_hidden_ex")
        return p, s, s8

    def sample_visible_class(self, P: ParamSet, h, seed=0, offset_v=0, offset_y=1, row0=0, want_p=False, want_s=False,
                             want_s8=True, want_py=False, want_y=True):
        """One launch for v ~ p(v|h) and y ~ p(y|h). -> (p_v, v_f32, v_u8, p_y, y1hot)"""
        h, hd = _state(h, self.device)
        B = h.shape[0]
        f = lambda n, dt: torch.empty(B, n, dtype=dt, device=self.device)
        pv = f(P.V, torch.float32) if want_p else None
        vs = f(P.V, torch.float32) if want_s else None
        v8 = f(P.V, torch.uint8) if want_s8 else None
        py = f(P.C, torch.float32) if want_py else None
        ys = f(P.C, torch.float32) if want_y else None
        ps = P.struct()
        check(self.lib.qbm_rbm_sample_visible_class_ex(C.byref(ps), ptr(h), hd, B, ptr(pv), ptr(vs), ptr(v8), ptr(py),
                                                       ptr(ys), seed, offset_v, offset_y, row0, current_stream()),
              "qbm_rbm_sample_visible_class_ex")
        return pv, vs, v8, py, ys

    def to_u8(self, x):
        x = _f32(x, self.device)
        y = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        check(self.lib.qbm_f32_to_u8(ptr(x), ptr(y), x.numel(), current_stream()), "qbm_f32_to_u8")
        return y

    def free_energy(self, P: ParamSet, x):
        """-> (-F [B,C], p(y|x) [B,C])"""
        x = _f32(x, self.device)
        B = x.shape[0]
        pre = torch.empty(B, P.H, dtype=torch.float32, device=self.device)
        nf = torch.empty(B, P.C, dtype=torch.float32, device=self.device)
        p = torch.empty(B, P.C, dtype=torch.float32, device=self.device)
        ps = P.struct()
        check(self.lib.qbm_rbm_free_energy(C.byref(ps), ptr(x), B, ptr(pre), ptr(nf), ptr(p), current_stream()),
              "qbm_rbm_free_energy")
        return nf, p

    # ---- training steps -----------------------------------------------------------------------
    def disc_step(self, P: ParamSet, G: GradSet, x, labels, apply_update, fuse, lr, factor, batch_size, sparse):
        """One discriminative_training pass (src/ClassificationRBM.py:101-140). -> p(y|x) [B,C]
        x: fp32, or uint8 {0,1} (binary inputs stay bytes: the uint8 kernels, same values)."""
        x, xd = _state(x, self.device)
        if not (torch.is_tensor(labels) and labels.dtype == torch.int32 and labels.device == self.device
                and labels.is_contiguous()):
            labels = torch.as_tensor(labels).to(device=self.device, dtype=torch.int32).contiguous()
        B = x.shape[0]
        key = (B, P.H, P.C)
        if self._disc_ws is None or self._disc_ws[0] != key:   # scratch reused from step to step (stream-ordered)
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
            bufs = (f(B, P.H), f(B, P.H), f(B, P.H), f(B, P.C), f(P.C, P.H))
            self._disc_ws = (key, bufs, DiscWorkspace(*(ptr(t) for t in bufs)))
        ws = self._disc_ws[2]
        p = torch.empty(B, P.C, dtype=torch.float32, device=self.device)
        ps, gs = P.struct(), G.struct(with_gb=False)
        check(self.lib.qbm_rbm_disc_step_ex(C.byref(ps), C.byref(gs), ptr(x), xd, ptr(labels), B, ptr(p), None, C.byref(ws),
                                            int(apply_update), int(fuse), lr, factor, batch_size, sparse,
                                            current_stream()), "qbm_rbm_disc_step_ex")
        return p

    def cd_step(self, P: ParamSet, G: GradSet, v0, y0, k, seed, offset, row0, chain_v, chain_y, fuse, lr, factor,
                batch_size, sparse, ws=None):
        """CD-k / PCD step: chain + gradient (+ fused update when `fuse`).
        v0: fp32 or uint8 {0,1} [B,V]; chain_v: uint8 [B,V] persistent state (PCD) or None.
        -> (ph0 f32, h u8, vk u8, yk f32, phk f32); with persistent chains (v_k, y_k) ARE chain_v / chain_y."""
        v0, vd = _state(v0, self.device)
        y0 = _f32(y0, self.device)
        B = v0.shape[0]
        if chain_v is not None and chain_v.dtype != torch.uint8:
            raise _lib.QbmHipError("persistent chain state must be uint8")
        if ws is None:   # scratch reused from step to step (stream-ordered), like the discriminative step's
            key = (B, P.V, P.H, P.C)
            if getattr(self, "_cd_ws", None) is None or self._cd_ws[0] != key:
                self._cd_ws = (key, self.cd_workspace(P, B))
            ws = self._cd_ws[1]
        cws = CdWorkspace(*(ptr(t) for t in ws))
        ps, gs = P.struct(), G.struct()
        check(self.lib.qbm_rbm_cd_step(C.byref(ps), C.byref(gs), ptr(v0), vd, ptr(y0), B, k, seed, offset, row0,
                                       ptr(chain_v), ptr(chain_y), C.byref(cws), int(fuse), lr, factor, batch_size,
                                       sparse, current_stream()), "qbm_rbm_cd_step")
        if chain_v is not None:   # PCD: the k-th iteration wrote straight into the chains
            return (ws[0], ws[1], chain_v, chain_y, ws[4])
        return ws

    def cd_workspace(self, P: ParamSet, B, fused=True):
        f = lambda n, dt: torch.empty(B, n, dtype=dt, device=self.device)
        sync = torch.zeros(_lib.QBM_CD_SYNC_WORDS, dtype=torch.int32, device=self.device) if fused else None
        return (f(P.H, torch.float32), f(P.H, torch.uint8), f(P.V, torch.uint8), f(P.C, torch.float32),
                f(P.H, torch.float32), sync)  # ph0, h, vk, yk, phk, state of the fused chain kernel

    def fused_chain_mode(self, mode=-1):
        """0 = never use the one-launch chain kernel, 1 = where it is faster (k >= 2; default), 2 = whenever the shape allows;
        -> previous mode (no argument: query).  Process-wide (include/qbm_hip.h: qbm_rbm_fused_chain_mode)."""
        return int(self.lib.qbm_rbm_fused_chain_mode(int(mode)))

    def check_chain_status(self, ws=None):
        """Raise if a fused chain launch that used this workspace (default: the one cd_step keeps) gave up
        (include/qbm_hip.h: qbm_rbm_cd_workspace.sync).  Synchronises; call it where the host synchronises anyway (end of an
        epoch, before results are read)."""
        if ws is None:
            ws = self._cd_ws[1] if getattr(self, "_cd_ws", None) is not None else ()
        if len(ws) > 5 and ws[5] is not None:
            st = int(ws[5][512].item())
            if st != 0:
                raise _lib.QbmHipError(
                    f"fused Gibbs chain kernel gave up (status {st}: " + {1: "a barrier between phases timed out", 2: "the workgroups of "
                    "a row block were not placed on one XCD"}.get(st, "unknown") + "): the results of that step are invalid; "
                    "set QBM_FUSED_CHAIN=0 to use one launch per phase")

    def update(self, P: ParamSet, G: GradSet, lr, factor, batch_size, sparse, with_gb=True):
        ps, gs = P.struct(), G.struct(with_gb=with_gb)
        check(self.lib.qbm_rbm_update(C.byref(ps), C.byref(gs), lr, factor, batch_size, sparse, current_stream()),
              "qbm_rbm_update")

    def two_phase_grad(self, V, H, Cn, A0, Y0, P0, A1, Y1, P1, G: GradSet):
        A0, d0 = _state(A0, self.device)
        A1, d1 = _state(A1, self.device)
        B = A0.shape[0]
        gs = G.struct()
        check(self.lib.qbm_rbm_two_phase_grad(V, H, Cn, B, ptr(A0), d0, ptr(Y0), ptr(P0), ptr(A1), d1, ptr(Y1), ptr(P1),
                                              C.byref(gs), None, None, 0.0, 0.0, 1.0, 0.0, current_stream()),
              "qbm_rbm_two_phase_grad")

    # ---- test hooks ---------------------------------------------------------------------------
    def math(self, which, x):
        x = _f32(x, self.device)
        y = torch.empty_like(x)
        check(self.lib.qbm_test_math({"exp": 0, "log": 1, "sigmoid": 2, "softplus": 3}[which], ptr(x), ptr(y),
                                     x.numel(), current_stream()), "qbm_test_math")
        return y

    def uniform(self, seed, offset, row0, rows, cols):
        out = torch.empty(rows, cols, dtype=torch.float32, device=self.device)
        check(self.lib.qbm_test_uniform(seed, offset, row0, rows, cols, ptr(out), current_stream()),
              "qbm_test_uniform")
        return out
