"""HIP Gibbs sampler behind the reference's two sampler boundaries:

  B2  `sample_Q(Q, num_reads) -> float32[num_reads, n]` -- the protocol of src/qubo/sampler.py:26-33
      (LocalSASampler / DWaveAdapter), consumed by src/train/pipeline.py:20,35;
  B3  the sampler dispatch of Disc_QBM.get_samples (src/model/faster_dqbm.py:571-684), used through
      HipQuboOps by disc_qbm.Disc_QBM(solver="HIP").

The annealer the reference calls (dwave-neal 0.5.9, absent here) is replaced by heat-bath sweeps on the
GPU (csrc/qubo_kernels.hip).  No CPU fallback: a missing library or GPU raises QbmHipError."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr


class HipQuboOps:
    def __init__(self, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.QbmHipError("no ROCm device visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device(device)

    def gibbs(self, lin, J, reads, sweeps, beta0=1.0, beta1=1.0, seed=0, offset=0, want_mean=True, chain0=0):
        """lin [B,n] f32, J [n,n] f32 symmetric zero-diagonal (device tensors)
        -> (states u8 [B,R,n], mean f64 [B,n]).  chain0: global index of the first chain (sharded minibatches)."""
        lin = lin.to(self.device, torch.float32).contiguous()
        J = J.to(self.device, torch.float32).contiguous()
        B, n = lin.shape
        if J.shape != (n, n):
            raise _lib.QbmHipError(f"J must be [{n},{n}], got {tuple(J.shape)}")
        states = torch.empty(B, reads, n, dtype=torch.uint8, device=self.device)
        mean = torch.empty(B, n, dtype=torch.float64, device=self.device) if want_mean else None
        check(self.lib.qbm_qubo_gibbs_ex(ptr(lin), ptr(J), B, reads, n, sweeps, beta0, beta1, seed, offset, int(chain0),
                                         ptr(states), ptr(mean), current_stream()), "qbm_qubo_gibbs_ex")
        return states, mean

    def gibbs_csr(self, lin, J, reads, sweeps, beta0=1.0, beta1=1.0, seed=0, offset=0, want_mean=True, chain0=0, cap=None):
        """gibbs() for SPARSE couplings too wide for the LDS triangle (n > 276): the non-zero entries of J as CSR rows in LDS,
        the chains' fields beside them (include/qbm_hip.h, "sparse couplings").  -> (states, mean, head): head = int32[2] on the
        device, (status, non-zero couplings found); status 1 = more than `cap` (default: what the LDS holds) and NOTHING was
        sampled -- the caller falls back to gibbs().  States are bit-identical to gibbs() on the same arguments."""
        lin = lin.to(self.device, torch.float32).contiguous()
        J = J.to(self.device, torch.float32).contiguous()
        B, n = lin.shape
        if J.shape != (n, n):
            raise _lib.QbmHipError(f"J must be [{n},{n}], got {tuple(J.shape)}")
        lim = int(self.lib.qbm_qubo_csr_max_entries(n, max(1, B * reads)))
        cap = lim if cap is None else int(cap)
        ws = torch.empty(int(self.lib.qbm_qubo_csr_workspace_bytes(n, cap)) // 4, dtype=torch.int32, device=self.device)
        states = torch.empty(B, reads, n, dtype=torch.uint8, device=self.device)
        mean = torch.empty(B, n, dtype=torch.float64, device=self.device) if want_mean else None
        check(self.lib.qbm_qubo_csr_build(ptr(J), n, cap, ptr(ws), current_stream()), "qbm_qubo_csr_build")
        check(self.lib.qbm_qubo_gibbs_csr(ptr(lin), ptr(ws), cap, B, reads, n, sweeps, None, beta0, beta1, seed, offset, int(chain0),
                                          ptr(states), ptr(mean), current_stream()), "qbm_qubo_gibbs_csr")
        return states, mean, ws[:2]

    # ---- device-resident Disc_QBM state: theta = [W_vh (1+D)*H | W_vo D | b_h H | b_o 1] float64 (include/qbm_hip.h) ----
    def linear_fields_dev(self, X, labels, theta, D, H, beta_eff=1.0):
        """x-dependent QUBO diagonal with the parameters read from the device buffer theta -> lin fp32 [B,n]."""
        B = X.shape[0]
        n = H if labels is not None else H + 1
        lin = torch.empty(B, n, dtype=torch.float32, device=self.device)
        check(self.lib.qbm_qbm_linear_fields_dev(ptr(X), ptr(labels), ptr(theta), B, D, H, float(beta_eff), ptr(lin),
                                                 current_stream()), "qbm_qbm_linear_fields_dev")
        return lin

    def couplings(self, theta, W_hh, H, clamped, beta_eff=1.0):
        """-> (J fp32 [n,n], rowabs f64 [n], rowmin f64 [n]); W_hh: float64 [H,H] device tensor or None (restricted)."""
        n = H if clamped else H + 1
        J = torch.empty(n, n, dtype=torch.float32, device=self.device)
        rowabs = torch.empty(n, dtype=torch.float64, device=self.device)
        rowmin = torch.empty(n, dtype=torch.float64, device=self.device)
        check(self.lib.qbm_qbm_couplings(ptr(theta), ptr(W_hh), H, int(bool(clamped)), float(beta_eff), ptr(J), ptr(rowabs),
                                         ptr(rowmin), current_stream()), "qbm_qbm_couplings")   # W_vh row 0 = theta[:H]
        return J, rowabs, rowmin

    def beta_range(self, lin, rowabs, rowmin):
        """Per-image annealing range -> beta fp32 [B,2] = (hot, cold), on the device."""
        B, n = lin.shape
        beta = torch.empty(B, 2, dtype=torch.float32, device=self.device)
        check(self.lib.qbm_qbm_beta_range(ptr(lin), ptr(rowabs), ptr(rowmin), B, n, ptr(beta), current_stream()),
              "qbm_qbm_beta_range")
        return beta

    def gibbs_dev(self, lin, J, reads, sweeps, beta, seed=0, offset=0, chain0=0):
        """Like gibbs(), with the per-image range beta[B,2] read from device memory."""
        B, n = lin.shape
        states = torch.empty(B, reads, n, dtype=torch.uint8, device=self.device)
        mean = torch.empty(B, n, dtype=torch.float64, device=self.device)
        check(self.lib.qbm_qubo_gibbs_dev(ptr(lin), ptr(J), B, reads, n, sweeps, ptr(beta), 1.0, 1.0, seed, offset, int(chain0),
                                          ptr(states), ptr(mean), current_stream()), "qbm_qubo_gibbs_dev")
        return states, mean

    def stats_diff(self, X, labels, mean_c, mean_u, D, H, out, nll=False):
        """out[theta layout] = statistics(clamped) - statistics(unclamped) of the images given; nll: out holds one more
        element, the sum of the images' NLL terms (faster_dqbm.py:986-991), from the same launch."""
        fn = self.lib.qbm_qbm_stats_diff_nll if nll else self.lib.qbm_qbm_stats_diff
        check(fn(ptr(X), ptr(labels), ptr(mean_c), ptr(mean_u), X.shape[0], D, H, ptr(out), current_stream()),
              "qbm_qbm_stats_diff_nll" if nll else "qbm_qbm_stats_diff")
        return out

    def apply_update(self, theta, diff, D, H, lr, batch):
        check(self.lib.qbm_qbm_apply_update(ptr(theta), ptr(diff), D, H, float(lr), float(batch), current_stream()),
              "qbm_qbm_apply_update")

    def linear_fields(self, X, labels, W_vh, W_vo, b_h, b_o, beta_eff=1.0):
        """x-dependent QUBO diagonal of Disc_QBM.create_qubo_matrix_from (faster_dqbm.py:225-284), f64 in,
        fp32 [B,n] out on the device.  labels None = unclamped (n = 1+H), else clamped (n = H)."""
        d64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device) \
            if not torch.is_tensor(a) else a.to(self.device, torch.float64).contiguous()
        X, W_vh, b_h = d64(X), d64(W_vh), d64(b_h)
        W_vo = d64(W_vo).reshape(-1)
        B, D = X.shape
        H = W_vh.shape[1]
        if W_vh.shape[0] != D + 1:
            raise _lib.QbmHipError("W_vh must have 1 + dim_input rows (row 0 = label row)")
        lab = None if labels is None else torch.as_tensor(np.asarray(labels)).to(self.device, torch.int32).contiguous()
        n = H if lab is not None else H + 1
        lin = torch.empty(B, n, dtype=torch.float32, device=self.device)
        b_o = float(np.asarray(b_o.cpu() if torch.is_tensor(b_o) else b_o).reshape(-1)[0])
        check(self.lib.qbm_qbm_linear_fields(ptr(X), ptr(lab), ptr(W_vh), ptr(W_vo), ptr(b_h), b_o, B, D, H,
                                             float(beta_eff), ptr(lin), current_stream()), "qbm_qbm_linear_fields")
        return lin


    def batch_stats(self, X, labels, mean, H):
        """get_average_configuration_batch (faster_dqbm.py:754-848) from per-image means (device, f64).
        -> (W_vh [(1+D),H], W_vo [D], b_h [H], b_o [1]) float64 device tensors"""
        X = X.to(self.device, torch.float64).contiguous()
        mean = mean.to(self.device, torch.float64).contiguous()
        B, D = X.shape
        lab = None if labels is None else torch.as_tensor(np.asarray(labels)).to(self.device, torch.int32).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float64, device=self.device)
        W_vh, W_vo, b_h, b_o = f(1 + D, H), f(D), f(H), f(1)
        check(self.lib.qbm_qbm_batch_stats(ptr(X), ptr(lab), ptr(mean), B, D, H, ptr(W_vh), ptr(W_vo), ptr(b_h),
                                           ptr(b_o), current_stream()), "qbm_qbm_batch_stats")
        return W_vh, W_vo, b_h, b_o


def auto_beta_range(lin: np.ndarray, J: np.ndarray):
    """Hot/cold inverse temperatures for an annealing schedule, from the spread of single-flip energy
    changes (the heuristic simulated annealers use: hot = half of the largest flips accepted, cold = the
    smallest flip accepted 1 time in 100).  lin [B,n] or [n], J [n,n] symmetric, zero diagonal."""
    lin = np.atleast_2d(np.asarray(lin, dtype=np.float64))
    J = np.asarray(J, dtype=np.float64)
    big = np.abs(lin).max(axis=0) + np.abs(J).sum(axis=1)          # largest |dE| a flip of spin i can cause
    coeffs = np.concatenate([np.abs(lin).ravel(), np.abs(J).ravel()])
    coeffs = coeffs[coeffs > 0]
    if coeffs.size == 0 or not np.isfinite(big.max()) or big.max() <= 0:
        return 0.1, 1.0
    hot = np.log(2.0) / big.max()
    cold = np.log(100.0) / coeffs.min()
    cold = min(max(cold, 10.0 * hot), 1e4 * hot)                     # keep the geometric schedule sane
    return float(hot), float(cold)


class HipGibbsSampler:
    """Drop-in for src/qubo/sampler.py::LocalSASampler (same constructor keywords, same sample_Q).

    The protocol hands over ONE host matrix per call (src/train/pipeline.py:20, 35), so what a call costs besides the sweeps
    is host work and PCIe.  Round 4: the matrix goes to the device AS IT IS through a pinned staging buffer the sampler keeps
    (one memcpy, one asynchronous copy), and everything numpy did per call -- the diagonal, (off + off.T).astype(float32), the
    annealing range -- happens there in two launches (qbm_qubo_from_Q, qbm_qbm_beta_range); the samples come back as float32
    through a second pinned buffer.  One synchronisation per call."""

    def __init__(self, num_sweeps: int = 1000, seed: int | None = None, beta_range=None, device="cuda:0"):
        self.ops = HipQuboOps(device)
        self.num_sweeps = int(num_sweeps)
        self.seed = 0 if seed is None else int(seed)
        self.beta_range = None if beta_range is None else (float(beta_range[0]), float(beta_range[1]))
        self._calls = 0
        self.sparse = True  # wide QUBOs (n > 276): try the CSR-in-LDS kernel first (False: always the dense kernel; same samples)
        self.last_kernel = None
        self._bufs = {}     # (n, dtype, reads) -> staging and device buffers, kept from call to call

    def _buffers(self, n, dtype, reads):
        key = (n, dtype, reads)
        b = self._bufs.get(key)
        if b is None:
            if len(self._bufs) >= 4:
                self._bufs.pop(next(iter(self._bufs)))
            dev = self.ops.device
            f = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
            b = self._bufs[key] = dict(
                q_pin=torch.empty(n * n, dtype=dtype).pin_memory(), q_dev=f(n * n, dt=dtype), lin=f(1, n), J=f(n, n),
                rowabs=f(n, dt=torch.float64), rowmin=f(n, dt=torch.float64), any_off=torch.zeros(1, dtype=torch.int32, device=dev),
                beta=f(1, 2), states=torch.empty(1, reads, n, dtype=torch.uint8, device=dev), out=f(reads, n),
                out_pin=torch.empty(reads, n, dtype=torch.float32).pin_memory(), flag_pin=torch.zeros(2, dtype=torch.int32).pin_memory(),
                csr=None, cap=0)
            if n > 276:   # (the triangle of J no longer fits a CU's LDS: qubo_kernels.hip, launch_gibbs)
                b["cap"] = int(self.ops.lib.qbm_qubo_csr_max_entries(n, reads))
                b["csr"] = torch.zeros(int(self.ops.lib.qbm_qubo_csr_workspace_bytes(n, b["cap"])) // 4, dtype=torch.int32, device=dev)
            if self.beta_range is not None:
                b["beta"].copy_(torch.tensor([self.beta_range], dtype=torch.float32))
            b["q_np"] = b["q_pin"].numpy().reshape(n, n)
        return b

    def sample_Q(self, Q: np.ndarray, num_reads: int) -> np.ndarray:
        """Q: upper-triangular QUBO (minimise s^T Q s) -> float32 [num_reads, n] in {0,1}, column j = variable j
        (src/qubo/sampler.py:26-33)."""
        Q = np.asarray(Q)
        if Q.dtype != np.float32:
            Q = np.asarray(Q, dtype=np.float64)
        if Q.ndim != 2 or Q.shape[0] != Q.shape[1]:
            raise ValueError("QUBO matrix must be square")
        n, reads = Q.shape[0], int(num_reads)
        f64 = Q.dtype == np.float64
        b = self._buffers(n, torch.float64 if f64 else torch.float32, reads)
        lib, st = self.ops.lib, current_stream()
        np.copyto(b["q_np"], Q)                                           # the call's one pass over the matrix on the host
        b["q_dev"].copy_(b["q_pin"], non_blocking=True)
        check(lib.qbm_qubo_from_Q(ptr(b["q_dev"]), int(f64), n, ptr(b["lin"]), ptr(b["J"]), ptr(b["rowabs"]), ptr(b["rowmin"]),
                                  ptr(b["any_off"]), st), "qbm_qubo_from_Q")
        if self.beta_range is None:   # hot / cold from the QUBO's coefficients, on the device (auto_beta_range is the host statement)
            check(lib.qbm_qbm_beta_range(ptr(b["lin"]), ptr(b["rowabs"]), ptr(b["rowmin"]), 1, n, ptr(b["beta"]), st), "qbm_qbm_beta_range")
        dense = lambda: check(lib.qbm_qubo_gibbs_dev(ptr(b["lin"]), ptr(b["J"]), 1, reads, n, self.num_sweeps, ptr(b["beta"]), 1.0, 1.0,
                                                     self.seed, self._calls << 20, 0, ptr(b["states"]), None, st), "qbm_qubo_gibbs_dev")
        sparse = b["csr"] is not None and self.sparse
        if sparse:
            # too wide for the LDS triangle: the non-zero couplings as CSR rows in LDS (a conv-QBM QUBO holds 2 % of them) -- the
            # same states bit for bit, at a fraction of a flip's cost.  Whether they fit is known on the device only: the status
            # word comes back with the samples, and a matrix that does not fit is sampled again by the dense kernel below.
            check(lib.qbm_qubo_csr_build(ptr(b["J"]), n, b["cap"], ptr(b["csr"]), st), "qbm_qubo_csr_build")
            check(lib.qbm_qubo_gibbs_csr(ptr(b["lin"]), ptr(b["csr"]), b["cap"], 1, reads, n, self.num_sweeps, ptr(b["beta"]), 1.0, 1.0,
                                         self.seed, self._calls << 20, 0, ptr(b["states"]), None, st), "qbm_qubo_gibbs_csr")
            b["flag_pin"][1:2].copy_(b["csr"][:1], non_blocking=True)
        else:
            dense()
        check(lib.qbm_u8_to_f32(ptr(b["states"]), ptr(b["out"]), reads * n, st), "qbm_u8_to_f32")
        b["out_pin"].copy_(b["out"], non_blocking=True)
        b["flag_pin"][0:1].copy_(b["any_off"], non_blocking=True)
        torch.cuda.current_stream().synchronize()                         # the call's one synchronisation
        self.last_kernel = "csr" if sparse else "dense"
        if sparse and int(b["flag_pin"][1]) != 0 and int(b["flag_pin"][0]) != 0:
            dense()                                                       # (more couplings than the LDS holds: the second and last try)
            check(lib.qbm_u8_to_f32(ptr(b["states"]), ptr(b["out"]), reads * n, st), "qbm_u8_to_f32")
            b["out_pin"].copy_(b["out"], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            self.last_kernel = "dense"
        self._calls += 1
        if int(b["flag_pin"][0]) == 0:
            # the reference short-circuits a QUBO without quadratic terms (src/qubo/sampler.py:13-17, 28-29):
            # s_i = 1 if h_i < 0, 0 if h_i > 0, a coin (one draw per variable, shared by all reads) if h_i == 0
            self._calls -= 1          # (no draws were consumed by the protocol's answer)
            h = np.diag(Q)
            coin = np.random.default_rng(self.seed).integers(0, 2, size=n)
            sol = np.where(h < 0, 1, np.where(h > 0, 0, coin)).astype(np.float32)
            return np.tile(sol, (reads, 1))
        return b["out_pin"].numpy().copy()
