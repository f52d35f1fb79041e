"""Disc_QBM -- drop-in mirror of the reference's src/model/faster_dqbm.py::Disc_QBM with the QUBO sampler
(simulated annealing in the third-party dwave-neal, or a D-Wave QPU) replaced by the batched HIP Gibbs
kernels (csrc/qubo_kernels.hip).  Boundary B3 of SURVEY.md 8.B.

What is mirrored (file:line of the reference in each docstring): constructor and parameter initialisation
order, QUBO construction, the statistics / update rule including its quirks (Q7 row
layout of the visible-to-hidden statistics, Q8 hidden-hidden weights never learn), NLL, predict, the epoch
loop with its per-epoch weight pickle and dump-on-exception.  What differs, deliberately:
  * `solver="HIP"` (default; "SA" is accepted as an alias): all images of a minibatch and all reads are
    sampled in ONE kernel launch per phase;
  * the learned parameters live ON THE DEVICE during training (one float64 buffer
    theta = [W_vh | W_vo | b_h | b_o], include/qbm_hip.h): fields, couplings, annealing range, sampling,
    statistics and the update all run there, nothing crosses PCIe per minibatch.  The public numpy attributes
    (`weights_all_visible_to_hidden`, ...) are views of the host copy, refreshed from the device when read and
    uploaded when assigned (modify them by assignment, not in place, while a model is training);
  * the annealing range is derived PER IMAGE from its QUBO's coefficients (qbm_qbm_beta_range), as the
    reference's annealer does per QUBO; `beta_range=(b, b)` gives plain Gibbs sampling at inverse temperature b;
  * the list-of-dicts sample format of the reference is only materialised by the compatibility methods
    get_samples / get_samples_batch;
  * no D-Wave token is read or printed (Q13), no process pool (Q10), `restricted=True` works (Q9);
  * QPU / embedding methods are out of scope.
There is no CPU fallback: sampling without the HIP library or a GPU raises QbmHipError."""
from __future__ import annotations

import os
import pickle
import random
from pathlib import Path

import numpy as np

from . import metrics


class Disc_QBM:
    # reference signature: src/model/faster_dqbm.py:58-61 (+ keyword-only extensions)
    def __init__(self, dim_input, num_classes, epochs=2, n_hidden_nodes=4, seed=77, solver="HIP", restricted=False,
                 sample_count=20, anneal_steps=20, beta_eff=1, param_string="", load_path="", speicherort=None,
                 parallelize=False, use_one_hot_encoding=False, use_old_parallization=True, *, beta_range=None,
                 device="cuda:0", ops=None, data_parallel=None):
        if use_one_hot_encoding:
            raise NotImplementedError("one-hot output nodes are not used by any reference driver (qbm_main.py:67)")
        self.epochs = epochs
        self.seed = seed
        self.training_history = metrics.History([], [], [], [], [], [], [])
        self.temperatures = None
        self.dim_input = dim_input
        self.n_hidden_nodes = n_hidden_nodes
        self.restricted = restricted
        self.parallelize = parallelize          # accepted for signature parity; the HIP path is always batched
        self.use_one_hot_encoding = use_one_hot_encoding
        self.n_output_nodes = 1
        self._theta = None                      # device master of the learned parameters (float64, flat)
        self._host_newer = True                 # the host copy has changes the device buffer lacks
        self._dev_newer = False                 # ... and the other way round
        self._whh_dev = None
        self._J_cache = {}
        # RNG order of the reference (:82-87): W_hh from the CURRENT global numpy state, then reseed and draw
        # W_vh, W_vo, W_oo, b_h, b_o
        if not restricted:
            self.weights_hidden_hidden = self.init_weights_hidden_hidden()
        else:
            self.weights_hidden_hidden = None
        (self._wvh, self._wvo, self.weights_output_output, self._bh, self._bo) = self.init_weights_for_supervised()
        self.param_string = param_string
        if solver not in ("HIP", "SA"):
            # the reference's message for an unknown solver string (:681-682)
            raise Exception('No valid solver specified. This build samples on the GPU: use solver="HIP" '
                            '("SA" is accepted as an alias); D-Wave / BMS solvers are out of scope.')
        self.solver_string = solver
        self.sa_time_used = 0
        self.anneal_steps = anneal_steps
        self.sample_count = sample_count
        self.beta_eff = beta_eff
        self.beta_range = beta_range            # None: per image, from its QUBO's coefficients (qbm_qbm_beta_range)
        self._device = device
        self._ops = ops
        self._draw_offset = 0
        # data parallelism: the images of a minibatch are split over the ranks of the initialised
        # torch.distributed group; every rank keeps a replica of the parameters
        self._dp = None if data_parallel is None else bool(data_parallel)
        if load_path != "":
            self.load_savepoint(load_path)
        self.speicherort = speicherort

    # ---- learned parameters: host copy <-> device master ------------------------------------------
    def _get(self, name):
        self._pull()
        return getattr(self, name)

    def _set(self, name, value):
        self._pull()
        setattr(self, name, np.asarray(value, dtype=np.float64))
        self._host_newer = True

    weights_all_visible_to_hidden = property(lambda s: s._get("_wvh"), lambda s, v: s._set("_wvh", v))          # [(1+D), H]
    weights_clamped_visible_to_output = property(lambda s: s._get("_wvo"), lambda s, v: s._set("_wvo", v))      # [D, 1]
    biases_hidden = property(lambda s: s._get("_bh"), lambda s, v: s._set("_bh", v))                            # [H]
    biases_output = property(lambda s: s._get("_bo"), lambda s, v: s._set("_bo", v))                            # [1]

    # W_hh never learns and beta_eff is a constructor constant, but both are public attributes of
    # the reference: assigning them drops what was derived from the old value (the uploaded W_hh, the cached couplings)
    @property
    def weights_hidden_hidden(self):
        return self._whh

    @weights_hidden_hidden.setter
    def weights_hidden_hidden(self, value):
        self._whh = value
        self._whh_dev = None
        self._J_cache = {}

    @property
    def beta_eff(self):
        return self._beta_eff

    @beta_eff.setter
    def beta_eff(self, value):
        self._beta_eff = value
        self._J_cache = {}

    @property
    def weight_objects(self):
        """The reference's pickle layout (:88-95), built from the CURRENT arrays (the reference builds the list once, so a
        savepoint loaded afterwards is not in it: ADVICE round 1)."""
        return [self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output, self.biases_hidden,
                self.biases_output, self.weights_output_output, self.weights_hidden_hidden]

    def _theta_len(self):
        D, H = self.dim_input, self.n_hidden_nodes
        return (1 + D) * H + D + H + 1

    def _push(self):
        """Host copy -> device buffer (when the host copy is the newer one)."""
        import torch

        if self._theta is not None and not self._host_newer:
            return
        flat = np.concatenate([self._wvh.reshape(-1), self._wvo.reshape(-1), self._bh.reshape(-1), self._bo.reshape(-1)])
        assert flat.size == self._theta_len()
        self._theta = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float64)).to(self.ops.device)
        self._host_newer, self._dev_newer = False, False
        self._J_cache.pop(False, None)        # the unclamped couplings hold the label row of W_vh

    def _pull(self):
        """Device buffer -> host copy, in place (when the device holds updates the host copy lacks)."""
        if not self._dev_newer:
            return
        self._dev_newer = False
        flat = self._theta.detach().cpu().numpy()
        D, H = self.dim_input, self.n_hidden_nodes
        o = 0
        for arr in (self._wvh, self._wvo, self._bh, self._bo):
            arr[...] = flat[o:o + arr.size].reshape(arr.shape)
            o += arr.size

    # ------------------------------------------------------------------------------------------
    @property
    def ops(self):
        if self._ops is None:
            from .sampler import HipQuboOps

            self._ops = HipQuboOps(self._device)   # raises without the HIP library / a GPU
        return self._ops

    def load_savepoint(self, savepoint):
        """src/model/faster_dqbm.py:169-190 (5- or 6-tuples)"""
        savepoint = Path(savepoint)
        if not savepoint.exists():
            raise FileNotFoundError("Savepoint file not found")
        with open(savepoint, "rb") as file:
            loaded = pickle.load(file)   # a file this class wrote (save_weights)
        assert len(loaded) in [5, 6]
        (self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output, self.biases_hidden,
         self.biases_output, self.weights_output_output) = loaded[:5]
        if len(loaded) == 6:
            self.weights_hidden_hidden = loaded[5]
            self._whh_dev = None
            self._J_cache.clear()

    def init_weights_for_supervised(self):
        """src/model/faster_dqbm.py:192-218"""
        random.seed(self.seed)
        np.random.seed(self.seed)
        w_vh = np.random.uniform(-1, 1, (self.n_output_nodes + self.dim_input, self.n_hidden_nodes))
        w_vo = np.random.uniform(-1, 1, (self.dim_input, self.n_output_nodes))
        w_oo = np.triu(np.random.uniform(-1, 1, (self.n_output_nodes, self.n_output_nodes)), k=1)
        b_h = np.random.uniform(-1, 1, self.n_hidden_nodes)
        b_o = np.random.uniform(-1, 1, self.n_output_nodes)
        return w_vh, w_vo, w_oo, b_h, b_o

    def init_weights_hidden_hidden(self):
        """src/model/faster_dqbm.py:220-223"""
        return np.triu(np.random.uniform(-1, 1, (self.n_hidden_nodes, self.n_hidden_nodes)), k=1)

    def create_qubo_matrix_from(self, input_vector, label=None):
        """src/model/faster_dqbm.py:225-284 (host numpy, for API parity and tests; the training path builds
        only the x-dependent diagonal, on the device)."""
        H, no = self.n_hidden_nodes, self.n_output_nodes
        w_vh, w_vo = self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output
        if label is not None:
            visible = np.concatenate((np.array([label]), input_vector), axis=0)
            q = np.diag(self.biases_hidden) + np.diag(np.matmul(visible, w_vh).flatten())
            if not self.restricted:
                q = q + self.weights_hidden_hidden
            return q / self.beta_eff
        upper = np.zeros((no + H, no + H))
        upper += np.pad(w_vh[0:no, :], ((0, H), (no, 0)), 'constant', constant_values=0)
        upper += np.pad(self.weights_output_output, ((0, H), (0, H)), 'constant', constant_values=0)
        if not self.restricted:
            upper[no:, no:] += self.weights_hidden_hidden
        biases = np.concatenate((self.biases_output, self.biases_hidden), axis=0)
        w_in = np.concatenate((w_vo, w_vh[no:, :]), axis=1)
        return (np.diag(biases) + np.diag(np.matmul(input_vector, w_in).flatten()) + upper) / self.beta_eff

    # ---- HIP sampling -------------------------------------------------------------------------
    def _couplings(self, clamped: bool) -> np.ndarray:
        """Image-independent symmetric couplings J (zero diagonal) of the QUBO (:242-243, :256-271), host statement
        (the training path builds the same matrix on the device: qbm_qbm_couplings)."""
        H = self.n_hidden_nodes
        n, o = (H, 0) if clamped else (H + 1, 1)
        J = np.zeros((n, n), np.float32)
        if not clamped:
            w = (self.weights_all_visible_to_hidden[0, :] / self.beta_eff).astype(np.float32)
            J[0, 1:] = w
            J[1:, 0] = w
        if not self.restricted:
            whh = np.triu(self.weights_hidden_hidden, k=1) / self.beta_eff
            J[o:, o:] += (whh + whh.T).astype(np.float32)
        return J

    def _couplings_dev(self, clamped: bool):
        """(J, rowabs, rowmin) on the device.  The clamped couplings are W_hh only: built once; the unclamped ones hold the
        label row of W_vh: rebuilt whenever theta has changed."""
        import torch

        self._push()
        hit = self._J_cache.get(clamped)
        if hit is not None:
            return hit
        if self._whh_dev is None and not self.restricted:
            self._whh_dev = torch.from_numpy(np.ascontiguousarray(self.weights_hidden_hidden, dtype=np.float64)).to(self.ops.device)
        out = self.ops.couplings(self._theta, self._whh_dev, self.n_hidden_nodes, clamped, self.beta_eff)
        self._J_cache[clamped] = out
        return out

    def _dp_group(self):
        """(rank, world) of the data-parallel group, (0, 1) when not distributed."""
        import torch.distributed as dist

        on = self._dp if self._dp is not None else (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if not on:
            return 0, 1
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("data_parallel=True needs an initialised torch.distributed process group")
        return dist.get_rank(), dist.get_world_size()

    @staticmethod
    def _shard(n_rows, rank, world):
        """Contiguous rows [lo, hi) of rank `rank`: ceil(n/world) rows each, the last ranks may get fewer (or none)."""
        per = (n_rows + world - 1) // world
        return min(rank * per, n_rows), min((rank + 1) * per, n_rows)

    def _to_dev(self, x_batch, y_batch):
        """(X float64 [B,D], labels int32 [B] or None) on the sampler's device; device tensors pass through."""
        import torch

        dev = self.ops.device
        if torch.is_tensor(x_batch):
            X = x_batch.to(dev, torch.float64).reshape(x_batch.shape[0], self.dim_input).contiguous()
        else:
            X = torch.from_numpy(np.ascontiguousarray(np.asarray(x_batch, dtype=np.float64).reshape(len(x_batch), self.dim_input))).to(dev)
        if y_batch is None:
            return X, None
        if torch.is_tensor(y_batch):
            return X, y_batch.to(dev, torch.int32).reshape(-1).contiguous()
        return X, torch.from_numpy(np.asarray(y_batch).reshape(-1).astype(np.int32)).to(dev)

    def sample_batch(self, x_batch, y_batch=None, rows=None):
        """Both the reference's get_samples_batch (:961-969) and the reduction over reads of
        get_average_configuration_batch (:777-780) for a whole minibatch, on the device:
        -> (states uint8 [B,R,n], mean float64 [B,n])  (device tensors).
        rows=(lo, hi): sample only these images of the minibatch (one rank's shard).  The annealing range is per image and
        the Philox rows are the global chain indices, so the shard's samples are exactly the rows [lo, hi) of the unsharded
        call."""
        import torch

        X, labels = self._to_dev(x_batch, y_batch)
        lo, hi = (0, X.shape[0]) if rows is None else rows
        offset = self._draw_offset
        self._draw_offset += int(self.anneal_steps) + 1
        clamped = labels is not None
        H = self.n_hidden_nodes
        n = H if clamped else H + 1
        if hi <= lo:
            dev = self.ops.device
            return (torch.empty(0, int(self.sample_count), n, dtype=torch.uint8, device=dev),
                    torch.empty(0, n, dtype=torch.float64, device=dev))
        self._push()
        Xs = X[lo:hi]
        ls = None if labels is None else labels[lo:hi]
        lin = self.ops.linear_fields_dev(Xs, ls, self._theta, self.dim_input, H, self.beta_eff)
        J, rowabs, rowmin = self._couplings_dev(clamped)
        if self.beta_range is not None:
            beta = torch.tensor([float(self.beta_range[0]), float(self.beta_range[1])], dtype=torch.float32,
                                device=self.ops.device).repeat(hi - lo, 1).contiguous()
        else:
            beta = self.ops.beta_range(lin, rowabs, rowmin)
        return self.ops.gibbs_dev(lin, J, int(self.sample_count), int(self.anneal_steps), beta, int(self.seed), offset,
                                  lo * int(self.sample_count))

    @staticmethod
    def _to_dicts(states_row):
        return [dict(enumerate(map(float, s))) for s in states_row]

    def get_samples(self, input_vector, label=None):
        """src/model/faster_dqbm.py:571-684 -> list (len R) of {variable index: 0.0/1.0}; unclamped variable
        order [out, h_0..h_{H-1}], clamped [h_0..]."""
        states, _ = self.sample_batch(np.asarray(input_vector)[None, :], None if label is None else [label])
        return self._to_dicts(states[0].cpu().numpy())

    def get_samples_batch(self, input_batch, label_batch=None):
        """src/model/faster_dqbm.py:961-969 -> list[list[dict]] (compatibility format; one launch)"""
        states, _ = self.sample_batch(input_batch, label_batch)
        st = states.cpu().numpy()
        return [self._to_dicts(st[i]) for i in range(st.shape[0])]

    def _stats(self, mean, x_batch, y_batch):
        import torch

        X = torch.from_numpy(np.ascontiguousarray(np.asarray(x_batch, dtype=np.float64).reshape(len(x_batch), self.dim_input)))
        labels = None if y_batch is None else np.asarray(y_batch).reshape(-1).astype(np.int32)
        W_vh, W_vo, b_h, b_o = self.ops.batch_stats(X, labels, mean, self.n_hidden_nodes)
        w_oo = np.zeros(self.weights_output_output.shape)
        w_hh = None if self.restricted else np.zeros(self.weights_hidden_hidden.shape)   # never filled (:831-845)
        return (b_h.cpu().numpy(), b_o.cpu().numpy(), W_vh.cpu().numpy(), W_vo.cpu().numpy().reshape(-1, 1), w_oo, w_hh)

    def get_average_configuration(self, samples, input_vector, label=None):
        """src/model/faster_dqbm.py:688-752, the statistics of ONE image from its list of sample dicts (host arithmetic: the
        training step uses the batched form below) -> (b_h, b_o, W_vh, W_vo, W_oo, W_hh).  Unlike the batched form this one
        does fill W_hh (upper triangle of the read-averaged products, :739-748) and keeps the reference's row order of W_vh
        (rows 0..D-1 = x, row D = the label when clamped: :712-724)."""
        no, H, D = self.n_output_nodes, self.n_hidden_nodes, self.dim_input
        S = np.vstack([np.array(list(s.values())) for s in samples]).astype(np.float64)          # [R, n]
        unclamped = label is None
        lab = None if unclamped else np.array(label).flatten()
        avg = np.average(S, axis=0)
        hid = S[:, no:] if unclamped else S
        b_h = avg[no:] if unclamped else avg
        b_o = avg[:no] if unclamped else lab
        x = np.asarray(input_vector, dtype=np.float64).reshape(-1)
        vis = x if unclamped else np.concatenate((x, lab))
        w_vh = np.zeros(self.weights_all_visible_to_hidden.shape)
        # np.average(x_v * samples[:, h]) entry by entry (:719-724): the product first, then the mean over the reads
        w_vh[:vis.shape[0], :] = np.average(vis[None, :, None] * hid[:, None, :], axis=0)
        w_vo = np.zeros(self.weights_clamped_visible_to_output.shape)
        out = S[:, :no] if unclamped else np.broadcast_to(lab.astype(np.float64), (1, no))
        w_vo[:D, :] = np.average(x[None, :, None] * out[:, None, :], axis=0)
        w_oo = np.zeros(self.weights_output_output.shape)
        for o in range(no):
            for o2 in range(o + 1, no):
                w_oo[o, o2] = np.average(S[:, o] * S[:, o2]) if unclamped else np.average(lab[o] * lab[o2])
        w_hh = None
        if not self.restricted:
            w_hh = np.triu(np.average(hid[:, :, None] * hid[:, None, :], axis=0), k=1)
        return b_h, b_o, w_vh, w_vo, w_oo, w_hh

    def get_best_combined_acc_auc(self):
        """src/model/faster_dqbm.py:1334-1339 -> (best epoch, its accuracy, its AUC, 0.5 acc + 0.5 AUC)"""
        h = self.training_history
        best = int(np.argmax(h.combined_acc_auc_per_epoch))
        return best, h.acc_per_epoch[best], h.auc_per_epoch[best], h.combined_acc_auc_per_epoch[best]

    def get_average_configuration_batch(self, samples_batch, x_batch, y_batch=None):
        """src/model/faster_dqbm.py:754-848 -> (b_h, b_o, W_vh, W_vo, W_oo, W_hh) statistics.
        samples_batch: the reference's list[list[dict]] OR a float64 [B,n] tensor of per-image means."""
        import torch

        if torch.is_tensor(samples_batch):
            mean = samples_batch
        else:
            mean = torch.from_numpy(np.stack([
                np.vstack([np.array(list(s.values())) for s in samples]).astype(np.float64).mean(axis=0)
                for samples in samples_batch]))
        return self._stats(mean, x_batch, y_batch)

    def compute_nll(self, y_batch, samples_unclamped, nll=None):
        """src/model/faster_dqbm.py:972-994: p1 = mean over reads of the output node, NLL of [1-p1, p1] in
        float32 with the reference's +1e-12.  samples_unclamped: list[list[dict]] or the [B,n] mean tensor."""
        import torch

        if torch.is_tensor(samples_unclamped):
            p1 = samples_unclamped[:, 0].cpu().numpy()
        else:
            p1 = np.array([np.mean([list(s.values())[0] for s in samples]) for samples in samples_unclamped])
        total = 0.0
        for i, y in enumerate(np.asarray(y_batch).reshape(-1)):
            probs = torch.tensor([1 - p1[i], p1[i]], dtype=torch.float32)
            log_probs = torch.log(probs + 1e-12)
            total += float(-log_probs[int(y)])
        return total / len(p1)

    def _step_device(self, X, labels, learning_rate):
        """One training iteration with every array on the device (src/model/faster_dqbm.py:998-1064):
        -> (sum of the output bias' error, sum of the NLL terms) as 0-dim float64 device tensors and the batch size they are to be
        divided by (on the host).  With a data-parallel group every rank
        is handed the SAME minibatch, samples only its rows, and the statistics difference plus the NLL sum meet in ONE
        float64 all-reduce; every replica then applies the same update."""
        import torch

        rank, world = self._dp_group()
        B = X.shape[0]
        lo, hi = self._shard(B, rank, world) if world > 1 else (0, B)
        _, mean_c = self.sample_batch(X, labels, rows=(lo, hi))      # positive / clamped phase (:1007)
        _, mean_u = self.sample_batch(X, rows=(lo, hi))              # negative / unclamped phase (:1011)
        D, H = self.dim_input, self.n_hidden_nodes
        buf = torch.zeros(self._theta_len() + 1, dtype=torch.float64, device=self.ops.device)   # [diff (theta layout) | NLL sum]
        if hi > lo:
            self.ops.stats_diff(X[lo:hi], labels[lo:hi], mean_c, mean_u, D, H, buf, nll=True)   # (the NLL sum from the same launch)
        if world > 1 or self._dp is True:   # (data_parallel=True on a one-rank group: the same code path, a one-rank all-reduce)
            import torch.distributed as dist

            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        self._push()
        self.ops.apply_update(self._theta, buf, D, H, learning_rate, float(B))   # theta -= lr * (diff / B) (:1023-1059)
        self._dev_newer = True
        self._J_cache.pop(False, None)                                            # the label row of W_vh has moved
        # W_oo (empty triu for one output node) and W_hh statistics are identically zero in the reference (Q8)
        # (the SUMS: their division by the batch size is the host's -- a float64 division by torch on the device is not correctly
        # rounded, and the error of the output bias is compared with the reference's numpy value bit for bit)
        return buf[-2], buf[-1], float(B)

    def train_for_one_iteration(self, x_batch, y_batch, learning_rate, nll=None):
        """src/model/faster_dqbm.py:998-1064 -> (errors_biases_output, avg_batch_nll)"""
        X, labels = self._to_dev(x_batch, y_batch)
        err_sum, nll_sum, Bf = self._step_device(X, labels, learning_rate)
        err_bo, avg_batch_loss = np.array([float(err_sum) / Bf]), float(nll_sum) / Bf
        self.training_history.nll_per_batch.append(avg_batch_loss)
        return err_bo, avg_batch_loss

    def split_into_batches(self, lst, batch_size):
        return [lst[i:i + batch_size] for i in range(0, len(lst), batch_size)]

    def save_weights(self, title, path="out"):
        """src/model/faster_dqbm.py:1069-1077 (same pickle layout, so old checkpoints interoperate)"""
        with open(f"{path}/{title}.pkl", "wb") as f:
            pickle.dump(self.weight_objects, f)

    def predict_batch(self, X):
        """predict (:1227-1241) for many images in one launch -> int array of rounded output means.  With a data-parallel
        group the images are split over the ranks and the predictions meet in one all-reduce."""
        import torch

        rank, world = self._dp_group()
        Xd, _ = self._to_dev(X, None)
        n = Xd.shape[0]
        if world <= 1:
            _, mean = self.sample_batch(Xd)
            p1 = mean[:, 0]
        else:
            import torch.distributed as dist

            lo, hi = self._shard(n, rank, world)
            _, mean = self.sample_batch(Xd, rows=(lo, hi))
            p1 = torch.zeros(n, dtype=torch.float64, device=self.ops.device)
            if hi > lo:
                p1[lo:hi] = mean[:, 0]
            dist.all_reduce(p1, op=dist.ReduceOp.SUM)
        return np.round(p1.cpu().numpy()).astype(int)   # numpy rounds half to even, like the reference

    def predict(self, data):
        """src/model/faster_dqbm.py:1227-1241 -> (rounded output, samples of the output node)"""
        states, mean = self.sample_batch(np.asarray(data)[None, :])
        out = states[0, :, 0:self.n_output_nodes].cpu().numpy().astype(np.float64)
        rounded = np.round(np.mean(out, axis=0)).astype(int)
        return rounded[0], out.flatten()

    def train_model(self, train_X, train_Y, val_X, val_Y, batch_size=8, learning_rate=0.005):
        """src/model/faster_dqbm.py:1079-1166.  The training and validation sets are moved to the device once; a minibatch is
        a slice of the resident set, and the per-batch error / NLL stay on the device until the end of the epoch (no host
        synchronisation per minibatch).  Files are written by rank 0 only."""
        rank, _ = self._dp_group()
        save_folder = (self.speicherort or "") + self.param_string
        if rank == 0:
            os.makedirs(save_folder, exist_ok=True)
        print("Training with \n" f"batch size: {batch_size}\n", f"learning rate: {learning_rate}\n",
              f"hidden nodes: {self.n_hidden_nodes}\n", f"sample count: {self.sample_count}\n",
              f"beta eff: {self.beta_eff}\n")
        Xd, Yd = self._to_dev(train_X, np.asarray(train_Y))
        Xv, _ = self._to_dev(val_X, None)
        n_train = Xd.shape[0]
        for epoch in range(1, self.epochs + 1):
            batchnum = 1
            num_batches = max(1, n_train // batch_size)
            errs, nlls, bsz = [], [], []
            import torch

            def flush():   # the epoch's ONE synchronisation: per-batch error / NLL to the host (what has been enqueued so far)
                e = [v / bf for v, bf in zip(torch.stack(errs).cpu().tolist(), bsz)] if errs else []
                n = [v / bf for v, bf in zip(torch.stack(nlls).cpu().tolist(), bsz)] if nlls else []
                return e, n

            try:
                for b in range(0, n_train, batch_size):
                    xb, yb = Xd[b:b + batch_size], Yd[b:b + batch_size]       # the last partial batch is kept (:1106-1111)
                    if xb.shape[0] == 0:
                        continue
                    err_sum, nll_sum, Bf = self._step_device(xb, yb, learning_rate)
                    errs.append(err_sum)
                    nlls.append(nll_sum)
                    bsz.append(Bf)
                    batchnum += 1
                errs_h, nlls_h = flush()      # inside the try: an asynchronous device failure surfaces HERE
            except Exception:
                # dump-on-exception, then re-raise (:1128-1132): weights, and the history with what this epoch has of it
                if rank == 0:
                    try:
                        e_part, n_part = flush()
                        self.training_history.errors_per_batch.extend(e_part)
                        self.training_history.nll_per_batch.extend(n_part)
                    except Exception:   # noqa: BLE001 -- the device is gone: dump what the host has
                        pass
                    try:
                        self.save_weights(f'e{epoch}_b{batchnum}_{self.param_string}', save_folder)
                    except Exception:   # noqa: BLE001
                        pass
                    metrics.save_history(f"{save_folder}/", self.training_history)
                raise
            self.training_history.errors_per_batch.extend(errs_h)             # np.mean of a 1-element array = the element
            self.training_history.nll_per_batch.extend(nlls_h)
            epoch_errors, epoch_nll = float(np.sum(errs_h)), float(np.sum(nlls_h))
            if rank == 0:
                self.save_weights(f'e{epoch}_{self.param_string}', save_folder)
            val_predictions = self.predict_batch(Xv)
            acc, _, _, _, auc = metrics.get_metrics(np.asarray(val_Y), val_predictions, ["0", "1"])
            self.training_history.acc_per_epoch.append(acc)
            self.training_history.auc_per_epoch.append(auc)
            self.training_history.combined_acc_auc_per_epoch.append(0.5 * acc + 0.5 * auc)
            self.training_history.error_per_epoch.append(epoch_errors / num_batches)
            self.training_history.nll_per_epoch.append(epoch_nll / num_batches)
        if rank == 0:
            for name, values in (("acc_per_epoch", self.training_history.acc_per_epoch),
                                 ("auc_per_epoch", self.training_history.auc_per_epoch),
                                 ("combined_acc_auc_per_epoch", self.training_history.combined_acc_auc_per_epoch)):
                with open(f"{save_folder}/{name}{self.seed}.pkl", "wb") as f:
                    pickle.dump(values, f)
