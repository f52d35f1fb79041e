"""Disc_QBM -- drop-in mirror of the reference's src/model/faster_dqbm.py::Disc_QBM with the QUBO sampler
(simulated annealing in the third-party dwave-neal, or a D-Wave QPU) replaced by the batched HIP Gibbs
kernels (csrc/qubo_kernels.hip).  Boundary B3 of SURVEY.md 8.B.

What is mirrored (file:line of the reference in each docstring): constructor and parameter initialisation
order, QUBO construction, the statistics / update rule including its quirks (Q7 row
layout of the visible-to-hidden statistics, Q8 hidden-hidden weights never learn), NLL, predict, the epoch
loop with its per-epoch weight pickle and dump-on-exception.  What differs, deliberately:
  * `solver="HIP"` (default; "SA" is accepted as an alias): all images of a minibatch and all reads are
    sampled in ONE kernel launch per phase, statistics are reduced on the device; the list-of-dicts sample
    format of the reference is only materialised by the compatibility methods get_samples /
    get_samples_batch;
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
        # RNG order of the reference (:82-87): W_hh from the CURRENT global numpy state, then reseed and draw
        # W_vh, W_vo, W_oo, b_h, b_o
        if not restricted:
            self.weights_hidden_hidden = self.init_weights_hidden_hidden()
        else:
            self.weights_hidden_hidden = None
        (self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output, self.weights_output_output,
         self.biases_hidden, self.biases_output) = self.init_weights_for_supervised()
        self.param_string = param_string
        self.weight_objects = [
            self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output,
            self.biases_hidden, self.biases_output,
            self.weights_output_output, self.weights_hidden_hidden if self.weights_hidden_hidden is not None else None
        ]
        if solver not in ("HIP", "SA"):
            # the reference's message for an unknown solver string (:681-682)
            raise Exception('No valid solver specified. This build samples on the GPU: use solver="HIP" '
                            '("SA" is accepted as an alias); D-Wave / BMS solvers are out of scope.')
        self.solver_string = solver
        self.sa_time_used = 0
        self.anneal_steps = anneal_steps
        self.sample_count = sample_count
        self.beta_eff = beta_eff
        self.beta_range = beta_range            # None: derived per minibatch from the QUBO coefficients
        self._device = device
        self._ops = ops
        self._draw_offset = 0
        # data parallelism: the images of a minibatch are split over the ranks of the initialised
        # torch.distributed group; every rank keeps a replica of the parameters
        self._dp = None if data_parallel is None else bool(data_parallel)
        if load_path != "":
            self.load_savepoint(load_path)
        self.speicherort = speicherort

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
        if len(loaded) == 6:
            (self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output, self.biases_hidden,
             self.biases_output, self.weights_output_output, self.weights_hidden_hidden) = loaded
        else:
            (self.weights_all_visible_to_hidden, self.weights_clamped_visible_to_output, self.biases_hidden,
             self.biases_output, self.weights_output_output) = loaded

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
        if label is not None:
            visible = np.concatenate((np.array([label]), input_vector), axis=0)
            q = np.diag(self.biases_hidden) + np.diag(np.matmul(visible, self.weights_all_visible_to_hidden).flatten())
            if not self.restricted:
                q = q + self.weights_hidden_hidden
            return q / self.beta_eff
        upper = np.zeros((no + H, no + H))
        upper += np.pad(self.weights_all_visible_to_hidden[0:no, :], ((0, H), (no, 0)), 'constant', constant_values=0)
        upper += np.pad(self.weights_output_output, ((0, H), (0, H)), 'constant', constant_values=0)
        if not self.restricted:
            upper[no:, no:] += self.weights_hidden_hidden
        biases = np.concatenate((self.biases_output, self.biases_hidden), axis=0)
        w_in = np.concatenate((self.weights_clamped_visible_to_output, self.weights_all_visible_to_hidden[no:, :]), axis=1)
        return (np.diag(biases) + np.diag(np.matmul(input_vector, w_in).flatten()) + upper) / self.beta_eff

    # ---- HIP sampling -------------------------------------------------------------------------
    def _couplings(self, clamped: bool) -> np.ndarray:
        """Image-independent symmetric couplings J (zero diagonal) of the QUBO (:242-243, :256-271)."""
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

    def sample_batch(self, x_batch, y_batch=None, rows=None):
        """Both the reference's get_samples_batch (:961-969) and the reduction over reads of
        get_average_configuration_batch (:777-780) for a whole minibatch, on the device:
        -> (states uint8 [B,R,n], mean float64 [B,n])  (device tensors).
        rows=(lo, hi): sample only these images of the minibatch (one rank's shard).  The annealing range still comes
        from the whole minibatch and the Philox rows are the global chain indices, so the shard's samples are exactly
        the rows [lo, hi) of the unsharded call."""
        import torch

        from .sampler import auto_beta_range

        X = np.ascontiguousarray(np.asarray(x_batch, dtype=np.float64).reshape(len(x_batch), self.dim_input))
        labels = None if y_batch is None else np.asarray(y_batch).reshape(-1).astype(np.int32)
        lin = self.ops.linear_fields(X, labels, self.weights_all_visible_to_hidden,
                                     self.weights_clamped_visible_to_output, self.biases_hidden, self.biases_output,
                                     self.beta_eff)
        J = self._couplings(clamped=labels is not None)
        if self.beta_range is not None:
            b0, b1 = self.beta_range
        else:
            b0, b1 = auto_beta_range(lin.cpu().numpy(), J)
        offset = self._draw_offset
        self._draw_offset += int(self.anneal_steps) + 1
        lo, hi = (0, len(X)) if rows is None else rows
        n = lin.shape[1]
        if hi <= lo:
            dev = lin.device
            return (torch.empty(0, int(self.sample_count), n, dtype=torch.uint8, device=dev),
                    torch.empty(0, n, dtype=torch.float64, device=dev))
        return self.ops.gibbs(lin[lo:hi].contiguous(), torch.from_numpy(J), int(self.sample_count), int(self.anneal_steps),
                              b0, b1, int(self.seed), offset, True, lo * int(self.sample_count))

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

    def train_for_one_iteration(self, x_batch, y_batch, learning_rate, nll=None):
        """src/model/faster_dqbm.py:998-1064 -> (errors_biases_output, avg_batch_nll)"""
        rank, world = self._dp_group()
        if world > 1:
            return self._train_for_one_iteration_dp(x_batch, y_batch, learning_rate, rank, world)
        _, mean_c = self.sample_batch(x_batch, y_batch)      # positive / clamped phase (:1007)
        _, mean_u = self.sample_batch(x_batch)               # negative / unclamped phase (:1011)
        bh_c, bo_c, wvh_c, wvo_c, woo_c, whh_c = self._stats(mean_c, x_batch, y_batch)
        bh_u, bo_u, wvh_u, wvo_u, woo_u, whh_u = self._stats(mean_u, x_batch, None)
        B = np.asarray(x_batch).shape[0]
        err_bo = (bo_c - bo_u) / B
        err_bh = (bh_c - bh_u) / B
        err_wvh = (wvh_c - wvh_u) / B
        err_wvo = (wvo_c - wvo_u) / B
        err_woo = (woo_c - woo_u) / B
        self.biases_output -= learning_rate * err_bo
        self.biases_hidden -= learning_rate * err_bh
        self.weights_all_visible_to_hidden -= learning_rate * err_wvh
        self.weights_clamped_visible_to_output -= learning_rate * err_wvo
        self.weights_output_output -= learning_rate * err_woo
        if not self.restricted:
            self.weights_hidden_hidden -= learning_rate * ((whh_c - whh_u) / B)   # zeros: W_hh never learns (Q8)
        avg_batch_loss = self.compute_nll(y_batch, mean_u)
        self.training_history.nll_per_batch.append(avg_batch_loss)
        return err_bo, avg_batch_loss

    def _train_for_one_iteration_dp(self, x_batch, y_batch, learning_rate, rank, world):
        """The same update with the images of the minibatch split over the ranks: every rank
        is handed the SAME minibatch, samples only its rows, and the statistics of both phases (differences of sums
        over images) plus the NLL sum meet in ONE float64 all-reduce; every replica then applies the same update."""
        import torch
        import torch.distributed as dist

        X = np.asarray(x_batch)
        Y = np.asarray(y_batch).reshape(-1)
        B = X.shape[0]
        lo, hi = self._shard(B, rank, world)
        _, mean_c = self.sample_batch(X, Y, rows=(lo, hi))
        _, mean_u = self.sample_batch(X, rows=(lo, hi))
        H, D = self.n_hidden_nodes, self.dim_input
        flat = np.zeros(H + 1 + (1 + D) * H + D + 1, np.float64)   # [b_h | b_o | W_vh | W_vo | nll sum]
        if hi > lo:
            bh_c, bo_c, wvh_c, wvo_c, _, _ = self._stats(mean_c, X[lo:hi], Y[lo:hi])
            bh_u, bo_u, wvh_u, wvo_u, _, _ = self._stats(mean_u, X[lo:hi], None)
            flat[:H] = bh_c - bh_u
            flat[H] = float(np.asarray(bo_c - bo_u).reshape(-1)[0])
            flat[H + 1:H + 1 + (1 + D) * H] = (wvh_c - wvh_u).reshape(-1)
            flat[H + 1 + (1 + D) * H:-1] = (wvo_c - wvo_u).reshape(-1)
            flat[-1] = self.compute_nll(Y[lo:hi], mean_u) * (hi - lo)
        dev = mean_c.device if dist.get_backend() != "gloo" else torch.device("cpu")
        t = torch.from_numpy(flat).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        flat = t.cpu().numpy()
        err_bh, err_bo = flat[:H] / B, np.array([flat[H] / B])
        err_wvh = flat[H + 1:H + 1 + (1 + D) * H].reshape(1 + D, H) / B
        err_wvo = flat[H + 1 + (1 + D) * H:-1].reshape(D, 1) / B
        self.biases_output -= learning_rate * err_bo
        self.biases_hidden -= learning_rate * err_bh
        self.weights_all_visible_to_hidden -= learning_rate * err_wvh
        self.weights_clamped_visible_to_output -= learning_rate * err_wvo
        # W_oo (empty triu for one output node) and W_hh statistics are identically zero in the reference (Q8)
        avg_batch_loss = float(flat[-1] / B)
        self.training_history.nll_per_batch.append(avg_batch_loss)
        return err_bo, avg_batch_loss

    def split_into_batches(self, lst, batch_size):
        return [lst[i:i + batch_size] for i in range(0, len(lst), batch_size)]

    def save_weights(self, title, path="out"):
        """src/model/faster_dqbm.py:1069-1077 (same pickle layout, so old checkpoints interoperate)"""
        with open(f"{path}/{title}.pkl", "wb") as f:
            pickle.dump(self.weight_objects, f)

    def predict_batch(self, X):
        """predict (:1227-1241) for many images in one launch -> int array of rounded output means"""
        _, mean = self.sample_batch(X)
        return np.round(mean[:, 0].cpu().numpy()).astype(int)   # numpy rounds half to even, like the reference

    def predict(self, data):
        """src/model/faster_dqbm.py:1227-1241 -> (rounded output, samples of the output node)"""
        states, mean = self.sample_batch(np.asarray(data)[None, :])
        out = states[0, :, 0:self.n_output_nodes].cpu().numpy().astype(np.float64)
        rounded = np.round(np.mean(out, axis=0)).astype(int)
        return rounded[0], out.flatten()

    def train_model(self, train_X, train_Y, val_X, val_Y, batch_size=8, learning_rate=0.005):
        """src/model/faster_dqbm.py:1079-1166"""
        save_folder = (self.speicherort or "") + self.param_string
        os.makedirs(save_folder, exist_ok=True)
        print("Training with \n" f"batch size: {batch_size}\n", f"learning rate: {learning_rate}\n",
              f"hidden nodes: {self.n_hidden_nodes}\n", f"sample count: {self.sample_count}\n",
              f"beta eff: {self.beta_eff}\n")
        train_Y = np.asarray(train_Y)
        for epoch in range(1, self.epochs + 1):
            batchnum = 1
            num_batches = max(1, len(train_X) // batch_size)
            epoch_errors = 0
            epoch_nll = 0
            for b in range(0, len(train_X), batch_size):
                x_batch = train_X[b:b + batch_size]       # the last partial batch is kept (:1106-1111)
                y_batch = train_Y[b:b + batch_size]
                if len(x_batch) == 0:
                    continue
                try:
                    err_bo, avg_nll = self.train_for_one_iteration(x_batch, y_batch, learning_rate)
                    avg_err = np.mean(err_bo)
                    epoch_errors += avg_err
                    epoch_nll += avg_nll
                    self.training_history.errors_per_batch.append(avg_err)
                    batchnum += 1
                except Exception:
                    # dump-on-exception, then re-raise (:1128-1132)
                    self.save_weights(f'e{epoch}_b{batchnum}_{self.param_string}', save_folder)
                    metrics.save_history(f"{save_folder}/", self.training_history)
                    raise
            self.save_weights(f'e{epoch}_{self.param_string}', save_folder)
            val_predictions = self.predict_batch(val_X)
            acc, _, _, _, auc = metrics.get_metrics(np.asarray(val_Y), val_predictions, ["0", "1"])
            self.training_history.acc_per_epoch.append(acc)
            self.training_history.auc_per_epoch.append(auc)
            self.training_history.combined_acc_auc_per_epoch.append(0.5 * acc + 0.5 * auc)
            self.training_history.error_per_epoch.append(epoch_errors / num_batches)
            self.training_history.nll_per_epoch.append(epoch_nll / num_batches)
        for name, values in (("acc_per_epoch", self.training_history.acc_per_epoch),
                             ("auc_per_epoch", self.training_history.auc_per_epoch),
                             ("combined_acc_auc_per_epoch", self.training_history.combined_acc_auc_per_epoch)):
            with open(f"{save_folder}/{name}{self.seed}.pkl", "wb") as f:
                pickle.dump(values, f)
