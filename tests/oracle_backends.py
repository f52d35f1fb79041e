"""Oracle-backed test doubles of the HIP backends (tests only).  They expose the same methods as
rbm_ops.HipRbmOps / sampler.HipQuboOps but compute on the CPU with oracle/qbm_oracle.c, so that the HOST
logic of the product (ClassificationRBM / Disc_QBM classes, data-parallel sharding, RNG offsets) can be
exercised without a GPU.  The product never imports this module."""
import numpy as np
import torch

from oracle import qbm_oracle as orc


def _np(t, dtype=np.float32):
    if t is None:
        return None
    if torch.is_tensor(t):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=dtype)


class OracleQuboOps:
    """Stand-in for sampler.HipQuboOps.  `fixed` = {(clamped: bool): states uint8 [B,R,n]} replaces the sampler
    (the reference's own tests would mock the annealer the same way)."""

    def __init__(self, fixed=None):
        self.device = torch.device("cpu")
        self.fixed = fixed

    def linear_fields(self, X, labels, W_vh, W_vo, b_h, b_o, beta_eff=1.0):
        return torch.from_numpy(orc.qbm_linear_fields(_np(X, np.float64), labels, W_vh, W_vo, b_h, b_o, beta_eff))

    def gibbs(self, lin, J, reads, sweeps, beta0=1.0, beta1=1.0, seed=0, offset=0, want_mean=True, chain0=0):
        n = lin.shape[1]
        if self.fixed is not None:
            st = self.fixed[n]
            return torch.from_numpy(st), torch.from_numpy(st.astype(np.float64).mean(axis=1))
        st, mean = orc.qubo_gibbs(_np(lin), _np(J), reads, sweeps, beta0, beta1, seed, offset, chain0)
        return torch.from_numpy(st), torch.from_numpy(mean)

    def batch_stats(self, X, labels, mean, H):
        return tuple(torch.from_numpy(a) for a in orc.qbm_batch_stats(_np(X, np.float64), labels, _np(mean, np.float64), H))

    # ---- device-resident parameter path (theta = [W_vh | W_vo | b_h | b_o], include/qbm_hip.h) on CPU tensors ----
    @staticmethod
    def _split(theta, D, H):
        t = _np(theta, np.float64)
        o = (1 + D) * H
        return t[:o].reshape(1 + D, H), t[o:o + D], t[o + D:o + D + H], t[o + D + H:o + D + H + 1]

    def linear_fields_dev(self, X, labels, theta, D, H, beta_eff=1.0):
        W_vh, W_vo, b_h, b_o = self._split(theta, D, H)
        lab = None if labels is None else _np(labels, np.int32)
        return torch.from_numpy(orc.qbm_linear_fields(_np(X, np.float64), lab, W_vh, W_vo, b_h, b_o, beta_eff))

    def couplings(self, theta, W_hh, H, clamped, beta_eff=1.0):
        w_label = _np(theta, np.float64)[:H].reshape(1, H)            # W_vh row 0
        J = orc.qbm_couplings(w_label, None if W_hh is None else _np(W_hh, np.float64), bool(clamped), beta_eff)
        rowabs, rowmin = orc.qbm_coupling_stats(J)
        return torch.from_numpy(J), torch.from_numpy(rowabs), torch.from_numpy(rowmin)

    def beta_range(self, lin, rowabs, rowmin):
        return torch.from_numpy(orc.qbm_beta_range(_np(lin), _np(rowabs, np.float64), _np(rowmin, np.float64)))

    def gibbs_dev(self, lin, J, reads, sweeps, beta, seed=0, offset=0, chain0=0):
        n = lin.shape[1]
        if self.fixed is not None:
            st = self.fixed[n]
            return torch.from_numpy(st), torch.from_numpy(st.astype(np.float64).mean(axis=1))
        lin, J, beta = _np(lin), _np(J), _np(beta)
        states = np.empty((lin.shape[0], reads, n), np.uint8)
        mean = np.empty((lin.shape[0], n), np.float64)
        for b in range(lin.shape[0]):                                 # per-image annealing range: one oracle call per image
            states[b:b + 1], mean[b:b + 1] = orc.qubo_gibbs(lin[b:b + 1], J, reads, sweeps, float(beta[b, 0]), float(beta[b, 1]),
                                                            seed, offset, chain0 + b * reads)
        return torch.from_numpy(states), torch.from_numpy(mean)

    def stats_diff(self, X, labels, mean_c, mean_u, D, H, out, nll=False):
        d = orc.qbm_stats_diff(_np(X, np.float64), _np(labels, np.int32), _np(mean_c, np.float64), _np(mean_u, np.float64), H)
        out[:d.size] = torch.from_numpy(d)
        if nll:   # compute_nll's per-image term, summed (src/model/faster_dqbm.py:986-991), in numpy as the reference states it
            p1 = _np(mean_u, np.float64)[:, 0]
            probs = np.stack([1.0 - p1, p1], axis=1).astype(np.float32)
            lp = np.log(probs + np.float32(1e-12))
            out[d.size] = float(np.sum(-lp[np.arange(len(p1)), _np(labels, np.int64)].astype(np.float64)))
        return out

    def apply_update(self, theta, diff, D, H, lr, batch):
        n = (1 + D) * H + D + H + 1
        t, d = theta.numpy(), _np(diff, np.float64)
        t[:n] = t[:n] - lr * (d[:n] / batch)


class OracleRbmOps:
    """Stand-in for rbm_ops.HipRbmOps (CPU tensors in and out)."""

    name = "oracle-test-double"

    def __init__(self):
        self.device = torch.device("cpu")

    @staticmethod
    def _P(P):
        return [_np(t) for t in (P.W, P.b, P.d, P.U, P.c)]

    def sample_hidden(self, P, v, y=None, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        W, b, d, U, c = self._P(P)
        p, s = orc.rbm_sample_hidden(_np(v), _np(y), W, d, U, seed, offset, row0, sample=want_s)
        return torch.from_numpy(p), (torch.from_numpy(s) if want_s else None)

    def sample_visible(self, P, h, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        W, b, d, U, c = self._P(P)
        p, s = orc.rbm_sample_visible(_np(h), W, b, seed, offset, row0, sample=want_s)
        return torch.from_numpy(p), (torch.from_numpy(s) if want_s else None)

    def sample_class(self, P, h, seed=0, offset=0, row0=0, want_p=True, want_s=False):
        W, b, d, U, c = self._P(P)
        p, s = orc.rbm_sample_class(_np(h), U, c, seed, offset, row0, sample=want_s)
        return torch.from_numpy(p), (torch.from_numpy(s) if want_s else None)

    def free_energy(self, P, x):
        W, b, d, U, c = self._P(P)
        _, nf, p = orc.rbm_free_energy(_np(x), W, d, U, c)
        return torch.from_numpy(nf), torch.from_numpy(p)

    def to_u8(self, x):
        return torch.as_tensor(x).to(torch.uint8)

    def uniform(self, seed, offset, row0, rows, cols):
        return torch.from_numpy(orc.philox_uniform(seed, offset, row0, rows, cols))

    def _store(self, G, gW, gb, gd, gU, gc):
        G.gW.copy_(torch.from_numpy(gW)); G.gd.copy_(torch.from_numpy(gd))
        G.gU.copy_(torch.from_numpy(gU)); G.gc.copy_(torch.from_numpy(gc))
        if gb is not None:
            G.gb.copy_(torch.from_numpy(gb))

    def update(self, P, G, lr, factor, batch_size, sparse, with_gb=True, count_dev=None, guard_chain=False):
        if count_dev is not None:      # qbm_rbm_update_dev: the batch size is read from the (all-reduced) row-count slot
            batch_size = float(count_dev.reshape(-1)[0])
            if not (batch_size > 0.0) or batch_size > 3.0e38:
                return
        W, b, d, U, c = self._P(P)
        orc.rbm_update(W, b, d, U, c, _np(G.gW), _np(G.gb) if with_gb else None, _np(G.gd), _np(G.gU), _np(G.gc),
                       lr, factor, batch_size, sparse)
        for t, a in zip((P.W, P.b, P.d, P.U, P.c), (W, b, d, U, c)):
            t.copy_(torch.from_numpy(a))

    @staticmethod
    def _zero(G, with_gb=True):
        # the C ABI's contract for a call with no rows (include/qbm_hip.h, qbm_rbm_cd_step / qbm_rbm_disc_step): the
        # gradient blocks handed in are zeroed, the parameters are left alone
        for t in (G.gW, G.gd, G.gU, G.gc) + ((G.gb,) if with_gb else ()):
            t.zero_()

    def disc_step(self, P, G, x, labels, apply_update, fuse, lr, factor, batch_size, sparse, with_count=False, loss_acc=None,
                  p_out=None):
        if with_count:                 # qbm_rbm_grads.n_rows: the gradient launch stores its row count
            G.count.fill_(float(len(x)))
        if len(x) == 0:
            self._zero(G, with_gb=False)
            return torch.empty(0, P.C)
        W, b, d, U, c = self._P(P)
        gW, gd, gU, gc, p, _, _, bl = orc.rbm_disc_grad(_np(x), _np(labels, np.int32), W, d, U, c, with_loss=True)
        if loss_acc is not None:       # qbm_rbm_disc_workspace.loss_acc: += (double)(fp32 minibatch mean)
            loss_acc += torch.from_numpy(bl.astype(np.float64))
        self._store(G, gW, None, gd, gU, gc)
        if apply_update:
            self.update(P, G, lr, factor, batch_size, sparse, with_gb=False)
        return torch.from_numpy(p)

    def cd_step(self, P, G, v0, y0, k, seed, offset, row0, chain_v, chain_y, fuse, lr, factor, batch_size, sparse,
                ws=None, with_count=False, replay=None, monitor=None, draw_base=None):
        if draw_base is not None:      # qbm_rbm_cd_workspace.draw_base: device words added to (offset, row0)
            offset, row0 = offset + int(draw_base[0]), (row0 + int(draw_base[1])) & 0xFFFFFFFF
        if with_count:
            G.count.fill_(float(len(v0)))
        if len(v0) == 0:
            self._zero(G)
            return tuple(torch.empty(0, n) for n in (P.H, P.H, P.V, P.C, P.H))
        W, b, d, U, c = self._P(P)
        if monitor is not None:        # qbm_rbm_cd_workspace.mon_*: p(y|x0) before the update, its loss terms booked
            _, _, p = orc.rbm_free_energy(_np(v0), W, d, U, c)
            monitor[1].add_(torch.from_numpy(orc.rbm_class_loss(p, _np(monitor[0], np.int32))[1].astype(np.float64)))
        cv = None if chain_v is None else _np(chain_v)
        cy = None if chain_y is None else _np(chain_y)
        ph0, vk, yk, phk = orc.rbm_cd_chain(_np(v0), _np(y0), W, b, d, U, c, k, seed, offset, row0, cv, cy)
        if chain_v is not None:
            chain_v.copy_(torch.from_numpy(cv.astype(np.uint8)))
            chain_y.copy_(torch.from_numpy(cy))
        self._store(G, *orc.rbm_two_phase_grad(_np(v0), _np(y0), ph0, vk, yk, phk))
        if fuse:
            self.update(P, G, lr, factor, batch_size, sparse)
        return tuple(torch.from_numpy(a) for a in (ph0, vk.astype(np.uint8), vk.astype(np.uint8), yk, phk))


class OracleDataOps:
    """CPU double of device_data.HipDataOps (binarise / pack_bits / permutation / gather through the oracle), so that
    DeviceDataset's host logic -- batch walk, rank sharding, epoch counter -- is covered without a GPU."""

    def __init__(self):
        import torch

        self.device = torch.device("cpu")

    def binarise(self, x, threshold=0.5):
        import torch

        return (x.to(torch.float32) > threshold).to(torch.uint8)

    def pack_bits(self, x_u8):
        import torch

        return torch.from_numpy(np.packbits(x_u8.numpy().astype(np.uint8), axis=1))

    def permutation(self, seed, epoch, n):
        import torch

        return torch.from_numpy(orc.data_shuffle_perm(seed, epoch, n).astype(np.int32))

    def gather(self, src, src_kind, idx, num_visible, dst_kind=0, labels=None, num_classes=0, want_onehot=False, out=None):
        import torch

        res = orc.data_gather(src.numpy(), idx.numpy(), None if labels is None else labels.numpy(), num_classes,
                              num_visible if src_kind == 2 else None, np.float32 if dst_kind == 0 else np.uint8)
        if labels is None:
            got = (torch.from_numpy(res), None, None)
        else:
            rows, lab, one = res
            got = (torch.from_numpy(rows), torch.from_numpy(lab), torch.from_numpy(one) if want_onehot else None)
        if out is None:
            return got
        for dst, src_t in zip(out, got):   # the dataset's epoch buffers (same storage in every epoch)
            if dst is not None:
                dst.copy_(src_t)
        return out
