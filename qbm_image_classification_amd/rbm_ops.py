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
        self._struct = None

    def struct(self) -> RbmParams:
        if self._struct is None:   # built once per ParamSet (the tensors of a ParamSet never change identity)
            self._struct = RbmParams(ptr(self.W), ptr(self.b), ptr(self.d), ptr(self.U), ptr(self.c), self.V, self.H, self.C)
        return self._struct

    def same_tensors(self, W, b, d, U, c) -> bool:
        return self.W is W and self.b is b and self.d is d and self.U is U and self.c is c


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
        self._structs = {}

    def struct(self, with_gW=True, with_gb=True, with_count=False) -> RbmGrads:
        """with_count: the gradient launch stores its row count in `count` (qbm_rbm_grads.n_rows): the data-parallel path."""
        key = (with_gW, with_gb, with_count)
        st = self._structs.get(key)
        if st is None:
            st = self._structs[key] = RbmGrads(ptr(self.gW) if with_gW else None, ptr(self.gb) if with_gb else None,
                                               ptr(self.gd), ptr(self.gU), ptr(self.gc), ptr(self.count) if with_count else None)
        return st


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
        self._disc_ws = None     # the discriminative step's scratch of the last call; _disc_wss: one per minibatch shape
        self._disc_wss = {}
        self._cd_ws = None       # (key, workspace) of the last cd_step that used a cached workspace
        self._cd_wss = {}        # minibatch shape -> workspace (they share the one sync buffer)
        self._sync = None        # state of the fused chain kernel (round flags, status word)
        self._pending = []       # (draw offset, replay) of the steps since the last look at the status word
        self.status_agree = None # data-parallel callers: (status, offset) -> the group's (collective; see ClassificationRBM)

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
    def disc_step(self, P: ParamSet, G: GradSet, x, labels, apply_update, fuse, lr, factor, batch_size, sparse,
                  with_count=False, loss_acc=None, p_out=None, guard_chain=False, replay=None):
        """One discriminative_training pass (src/ClassificationRBM.py:101-140). -> p(y|x) [B,C]
        x: fp32, or uint8 {0,1} (binary inputs stay bytes: the uint8 kernels, same values).
        with_count: the gradient launch also stores its row count in G.count (the data-parallel path all-reduces it).
        loss_acc: float64 [2] device tensor; the step adds the minibatch's CrossEntropyLoss-on-probabilities (:142) and
        -log(p_y + 1e-8) mean (:188-191) to it -- the training loop's bookkeeping without a launch or a sync of its own.
        p_out: where p(y|x) goes (default: a fresh tensor; `False`: a scratch buffer nobody reads).
        guard_chain (hybrid training: discriminative and CD steps alternate on the same parameters): the step reads the status
        word of the fused chain launches (include/qbm_hip.h: qbm_rbm_disc_workspace.status) -- behind a CD step that gave up it
        stores nothing -- and is recorded with the pending CD steps, so that check_chain_status repeats it IN ORDER (`replay`:
        what to call for that; default: this call again)."""
        x, xd = _state(x, self.device)
        if not (torch.is_tensor(labels) and labels.dtype == torch.int32 and labels.device == self.device
                and labels.is_contiguous()):
            labels = torch.as_tensor(labels).to(device=self.device, dtype=torch.int32).contiguous()
        B = x.shape[0]
        key = (B, P.H, P.C, 0 if loss_acc is None else loss_acc.data_ptr())
        if self._disc_ws is None or self._disc_ws[0] != key:   # scratch reused from step to step (stream-ordered)
            hit = self._disc_wss.get(key)   # (one per minibatch shape: the full minibatch and a short last one alternate)
            if hit is None:
                if len(self._disc_wss) >= self.kWorkspaces:
                    self._disc_wss.pop(next(iter(self._disc_wss)))
                f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.device)
                bufs = (f(B, P.H), f(B, P.H), f(B, 2), f(B, P.C))
                hit = self._disc_wss[key] = (key, bufs, DiscWorkspace(ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(loss_acc)), loss_acc)
            self._disc_ws = hit
        ws = self._disc_ws[2]
        guard = bool(guard_chain) and B > 0
        if guard and len(self._pending) >= self.kPendingMax:   # between two steps: every recorded step is complete
            self.check_chain_status()
        ws.status = C.c_void_p(self._sync_buffer().data_ptr() + 4 * 512) if guard else None
        if p_out is None:
            p_out = torch.empty(B, P.C, dtype=torch.float32, device=self.device)
        elif p_out is False:
            p_out = self._disc_ws[1][3]
        check(self.lib.qbm_rbm_disc_step_ex(C.byref(P.struct()), C.byref(G.struct(with_gb=False, with_count=with_count)), ptr(x), xd,
                                            ptr(labels), B, ptr(p_out), None, C.byref(ws), int(apply_update), int(fuse), lr, factor,
                                            batch_size, sparse, current_stream()), "qbm_rbm_disc_step_ex")
        if guard:
            args = (P, G, x, labels, apply_update, fuse, lr, factor, batch_size, sparse)
            self._pending.append((None, replay if replay is not None else
                                  (lambda: self.disc_step(*args, with_count=with_count, loss_acc=loss_acc, p_out=p_out)), None))
        return p_out

    def cd_step(self, P: ParamSet, G: GradSet, v0, y0, k, seed, offset, row0, chain_v, chain_y, fuse, lr, factor,
                batch_size, sparse, ws=None, with_count=False, replay=None, monitor=None, draw_base=None):
        """CD-k / PCD step: chain + gradient (+ fused update when `fuse`).
        monitor: (labels int32 [B] device tensor, loss_acc float64 [2] device tensor) -- the training loop's bookkeeping: the
        step also evaluates p(y|x0) under the parameters before its update and adds the minibatch's two loss terms to
        loss_acc (include/qbm_hip.h: qbm_rbm_cd_workspace.mon_*); the workspaces cd_step keeps only.
        v0: fp32 or uint8 {0,1} [B,V]; chain_v: uint8 [B,V] persistent state (PCD) or None.
        with_count: the gradient launch also stores its row count in G.count (the data-parallel path all-reduces it).
        replay: what to call to repeat this step should its one-launch chain give up (default: this call again; the
        data-parallel caller passes chain + all-reduce + update).
        draw_base: int64 / uint64 [2] device tensor added on the device to (offset, row0) by every sampling launch
        (include/qbm_hip.h: qbm_rbm_cd_workspace.draw_base): a loop that captures its steps in a hipGraph passes offsets relative
        to the epoch and moves the two words between replays; the workspaces cd_step keeps only.
        -> (ph0 f32, h u8, vk u8, yk f32, phk f32); with persistent chains (v_k, y_k) ARE chain_v / chain_y."""
        v0, vd = _state(v0, self.device)
        y0 = _f32(y0, self.device)
        B = v0.shape[0]
        if chain_v is not None and chain_v.dtype != torch.uint8:
            raise _lib.QbmHipError("persistent chain state must be uint8")
        own_ws = ws is None
        if own_ws:   # scratch reused from step to step (stream-ordered), like the discriminative step's
            if len(self._pending) >= self.kPendingMax:   # between two steps: every recorded step is complete
                self.check_chain_status()
            ws = self._cd_workspace_cached(P, B)
        cws = ws[6] if len(ws) > 6 else CdWorkspace(*(ptr(t) for t in ws[:6]))
        if own_ws:
            if monitor is not None and B > 0:
                if ws[7][0] is None:   # x.W + d, p(y|x), per-row loss terms: made on first use
                    f = lambda n: torch.empty(B, n, dtype=torch.float32, device=self.device)
                    ws[7][:] = [f(P.H), f(P.C), f(2)]
                    cws.mon_pre, cws.mon_p, cws.mon_rowloss = (ptr(t) for t in ws[7])
                cws.mon_labels, cws.mon_loss_acc = ptr(monitor[0]), ptr(monitor[1])
            else:
                cws.mon_labels = None
            cws.draw_base = ptr(draw_base) if draw_base is not None else None
        check(self.lib.qbm_rbm_cd_step(C.byref(P.struct()), C.byref(G.struct(with_count=with_count)), ptr(v0), vd, ptr(y0), B, k,
                                       seed, offset, row0, ptr(chain_v), ptr(chain_y), C.byref(cws), int(fuse), lr, factor,
                                       batch_size, sparse, current_stream()), "qbm_rbm_cd_step")
        if own_ws and ws[5] is not None and draw_base is None and self._may_fuse(k, vd):
            # a launch that synchronises workgroups inside an XCD may have run (the one-launch chain, or the fused tail: last
            # hidden phase + gradient + update): keep what it takes to repeat the step on the launch-per-phase path -- and, for
            # a fused tail with the update inside, to FINISH it (qbm_rbm_cd_step_finish: part of the update may have been
            # applied) -- until the next look at the status word (check_chain_status: end of an epoch, or every kPendingMax steps)
            args = (P, G, v0, y0, k, seed, offset, row0, chain_v, chain_y, fuse, lr, factor, batch_size, sparse)
            pst, gst = P.struct(), G.struct(with_count=with_count)
            keep_ws = ws   # (the closure holds the workspace's TENSORS, not only `cws`' raw pointers: the cache may evict the
                           #  workspace of a short minibatch before check_chain_status runs)

            def finish(rnd, _keep=keep_ws):
                check(self.lib.qbm_rbm_cd_step_finish(C.byref(pst), C.byref(gst), ptr(v0), vd, ptr(y0), B, k, seed, offset, row0,
                                                      ptr(chain_v), ptr(chain_y), C.byref(cws), int(fuse), lr, factor, batch_size,
                                                      sparse, rnd, current_stream()), "qbm_rbm_cd_step_finish")

            self._pending.append((int(offset) & 0xFFFFFFFF, replay if replay is not None else
                                  (lambda: self.cd_step(*args, with_count=with_count, monitor=monitor)),
                                  finish if fuse else None))
        if chain_v is not None:   # PCD: the k-th iteration wrote straight into the chains
            return (ws[0], ws[1], chain_v, chain_y, ws[4])
        return ws[:6]

    kPendingMax = 64    # steps between two looks at the fused chain's status word (each look synchronises the stream)
    kWorkspaces = 8     # cached CD workspaces (one per minibatch shape: the full minibatch and a short last one)

    def _may_fuse(self, k, v_dtype):
        """Could qbm_rbm_cd_step have taken the one-launch chain for this call?  Conservative (the C side decides from the
        shape as well) and independent of the row count, so that the ranks of a data-parallel group -- whose shards of a short
        last minibatch differ -- record the same steps and look at the status word at the same steps."""
        mode = int(self.lib.qbm_rbm_fused_chain_mode(-1))   # process-wide switch: asked, not cached
        # (modes 1, 3 from k = 2 on and mode 2: the one-launch chain; mode 3 otherwise: the fused tail of the step)
        return v_dtype == QBM_U8 and (mode >= 2 or (mode == 1 and k >= 2))

    def _sync_buffer(self):
        """ONE state buffer of the fused chain kernel per HipRbmOps (= per stream of work): it outlives the workspaces, which
        are re-made when the minibatch shape changes (a short last minibatch), so a status word set in any step of an epoch
        is still there when check_chain_status looks (a per-workspace buffer was dropped with its
        workspace)."""
        if self._sync is None:
            self._sync = torch.zeros(_lib.QBM_CD_SYNC_WORDS, dtype=torch.int32, device=self.device)
        return self._sync

    def _cd_workspace_cached(self, P: ParamSet, B):
        key = (B, P.V, P.H, P.C)
        ws = self._cd_wss.get(key)
        if ws is None:
            if len(self._cd_wss) >= self.kWorkspaces:
                self._cd_wss.pop(next(iter(self._cd_wss)))
            t = self.cd_workspace(P, B, fused=False)[:5] + (self._sync_buffer(),)
            ws = self._cd_wss[key] = t + (CdWorkspace(*(ptr(x) for x in t)), [None, None, None])   # + struct, monitor buffers
        self._cd_ws = (key, ws)
        return ws

    def cd_workspace(self, P: ParamSet, B, fused=True):
        f = lambda n, dt: torch.empty(B, n, dtype=dt, device=self.device)
        sync = torch.zeros(_lib.QBM_CD_SYNC_WORDS, dtype=torch.int32, device=self.device) if fused else None
        return (f(P.H, torch.float32), f(P.H, torch.uint8), f(P.V, torch.uint8), f(P.C, torch.float32),
                f(P.H, torch.float32), sync)  # ph0, h, vk, yk, phk, state of the fused chain kernel

    def fused_chain_mode(self, mode=-1):
        """0 = never use the one-launch chain kernel, 1 = where it is faster (k >= 2; default), 2 = whenever the shape allows,
        3 = 1 + the fused tail of a step (opt-in: slower); -> previous mode (no argument: query).  Process-wide
        (include/qbm_hip.h: qbm_rbm_fused_chain_mode)."""
        return int(self.lib.qbm_rbm_fused_chain_mode(int(mode)))

    _STATUS_TEXT = {1: "a barrier between phases timed out", 2: "the workgroups of a row block were not placed on one XCD",
                    3: "the barrier between the last hidden phase and the gradient timed out",
                    4: "the workgroups of a column group were not placed on one XCD"}

    def check_chain_status(self, ws=None):
        """Look at the status word of the fused chain kernel (include/qbm_hip.h: qbm_rbm_cd_workspace.sync).  Synchronises;
        train_rbm calls it at the end of every epoch, cd_step every kPendingMax steps.
        ws given (a caller-owned workspace): raise if a launch on it gave up.
        ws None (the workspaces cd_step keeps): every launch behind one that gave up was a no-op (FAIL-SAFE in the header).
        A one-launch chain that gave up (status 1 / 2) has changed NOTHING: the steps since then are repeated here on the
        launch-per-phase path (same draws, same bits).  A fused tail that gave up (status 3 / 4) with the update inside may
        have applied the update of some column groups: that step is first FINISHED from its untouched workspace
        (qbm_rbm_cd_step_finish), then the later steps are repeated.  The process stays on the launch-per-phase path
        (qbm_rbm_fused_chain_mode(0)) and a RuntimeWarning says so.
        Steps of a shape the fused launches do not take (a short last minibatch) and the discriminative steps of hybrid
        training read the same status word (include/qbm_hip.h: FAIL-SAFE), were no-ops alike and are repeated in order.
        ONE case is not a bit-identical repeat: persistent chains (PCD) in a data-parallel group -- the ranks whose own launch
        was fine have advanced their chains in the step another rank lost (only its UPDATE was skipped, through the -inf row
        count) and in the steps after it, and the group's repeat starts from those chains: the parameters see the same
        estimator on chains a few Gibbs steps further on.  (A snapshot of the chains per step would cost every step a copy.)"""
        if ws is not None:
            if len(ws) > 5 and ws[5] is not None:
                st = int(ws[5][512].item())
                if st != 0:
                    raise _lib.QbmHipError(
                        f"fused Gibbs chain kernel gave up (status {st}: {self._STATUS_TEXT.get(st, 'unknown')}): " +
                        ("the step did not reach the parameters; re-zero the workspace's sync buffer and repeat it with "
                         "QBM_FUSED_CHAIN=0 (one launch per phase)" if st in (1, 2) else
                         "with fuse_update part of the step's update may have been applied: complete it with "
                         "qbm_rbm_cd_step_finish (word 515 = round) before re-zeroing the sync buffer; later steps did nothing"))
            return
        pending, self._pending = self._pending, []
        if self._sync is None or not pending:
            return
        st, off_lo, _, rnd = (int(x) & 0xFFFFFFFF for x in self._sync[512:516].tolist())   # one synchronising copy
        if self.status_agree is not None:   # data-parallel group: every rank repeats from the first step ANY rank lost
            st, off_lo = self.status_agree(st, off_lo)
        if st == 0:
            return
        import warnings

        first = next((i for i, (o, _, _) in enumerate(pending) if o == (off_lo & 0xFFFFFFFF)), None)
        if first is None:
            raise _lib.QbmHipError(
                f"fused Gibbs chain kernel gave up (status {st}: {self._STATUS_TEXT.get(st, 'unknown')}) in a step this object "
                "no longer holds: the steps since then did not reach the parameters; set QBM_FUSED_CHAIN=0 and repeat them")
        self.fused_chain_mode(0)
        finish = pending[first][2] if st in (3, 4) else None
        if finish is not None:   # the fused tail with the update inside: complete that step from its workspace, as it left it
            finish(rnd)
            first += 1
        self._sync.zero_()
        warnings.warn(f"fused Gibbs chain kernel gave up (status {st}: {self._STATUS_TEXT.get(st, 'unknown')}); nothing was "
                      f"written since. " + ("Finished that step from its workspace; r" if finish is not None else "R") +
                      f"epeating {len(pending) - first} step(s) with one launch per phase (same draws; same bits, except that "
                      "the persistent chains of a data-parallel group's healthy ranks have moved on) and "
                      "staying on that path (qbm_rbm_fused_chain_mode(0)).", RuntimeWarning)
        for _, fn, _ in pending[first:]:
            fn()

    def update(self, P: ParamSet, G: GradSet, lr, factor, batch_size, sparse, with_gb=True, count_dev=None, guard_chain=False):
        """count_dev: 1-float device tensor holding the batch size (the all-reduced G.count of the data-parallel step: no host
        round trip); guard_chain: no-op while the fused chain's status word is set (the step is repeated by
        check_chain_status)."""
        status = None
        if guard_chain and self._sync is not None:
            status = C.c_void_p(self._sync.data_ptr() + 4 * 512)
        check(self.lib.qbm_rbm_update_dev(C.byref(P.struct()), C.byref(G.struct(with_gb=with_gb)), lr, factor,
                                          float(batch_size) if count_dev is None else 1.0, ptr(count_dev), sparse, status,
                                          current_stream()), "qbm_rbm_update_dev")

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
