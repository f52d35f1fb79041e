"""ClassificationRBM -- drop-in mirror of the reference's src/ClassificationRBM.py on the HIP path.

Same constructor, public attributes and methods as the reference class (boundary B1);
every method that does arithmetic forwards to libqbm_hip.so through rbm_ops.HipRbmOps.  Differences,
all deliberate:
  * the parameters live on the ROCm device (the reference forces CPU: src/ClassificationRBM.py:21);
  * `train_rbm(method=...)` also accepts 'generative' and 'hybrid', which the reference declares
    (rbm_main.py:22-28) but rejects (src/ClassificationRBM.py:181-182): CD-k / PCD as specified in
    SURVEY.md 8.A-7, built from the reference's own conditionals (:43-60) and update rule (:88-99);
  * with torch.distributed initialised (one process per GPU, RCCL) a step all-reduces one flat
    gradient buffer and divides by the global batch size.
Reference quirks are mirrored on purpose: the loss is CrossEntropyLoss applied
to probabilities (Q1, :142); roc_auc_score is computed on hard predictions (Q4, :280).
"""
from __future__ import annotations

import functools
import gc
import os

import numpy as np
import torch

from .rbm_ops import GradSet, HipRbmOps, ParamSet


def _dist_ready():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class ClassificationRBM:
    # reference signature: src/ClassificationRBM.py:12-13 (+ keyword-only extensions)
    def __init__(self, num_visible, num_hidden, k, num_classes=2, learning_rate=0.05, sparse_constant=0.00,
                 use_cuda=False, seed=42, *, persistent=False, device=None, ops=None, data_parallel=None,
                 keep_grads=False):
        np.random.seed(seed)
        torch.manual_seed(seed)
        self.seed = seed
        self.num_visible = num_visible
        self.num_hidden = num_hidden
        self.k = k
        self.learning_rate = learning_rate
        self.num_classes = num_classes
        self.loss = torch.nn.CrossEntropyLoss()
        self.sparse_constant = sparse_constant
        self.persistent = persistent
        # keep_grads=True materialises weights_grad like the reference; False fuses the W update into the
        # gradient pass (single-GPU path only; the data-parallel path always materialises for the all-reduce)
        self.keep_grads = keep_grads

        if ops is None:
            if device is None:
                device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0"
            ops = HipRbmOps(device)  # raises if the HIP library or the GPU is missing: no CPU fallback
        self.ops = ops
        self.device = ops.device
        self.use_cuda = self.device.type == "cuda"

        # same draw order as the reference (:26-30): one torch.randn(V, H) right after manual_seed
        self.weights = (torch.randn(num_visible, num_hidden) * 0.1).to(self.device)
        self.visible_bias = (torch.ones(num_visible) * 0.5).to(self.device)
        self.hidden_bias = torch.zeros(num_hidden).to(self.device)
        self.class_bias = torch.zeros(num_classes).to(self.device)
        self.class_weights = (torch.zeros(num_classes, num_hidden) * 0.5).to(self.device)

        self._grads = GradSet(num_visible, num_hidden, num_classes, self.device)
        self.weights_grad = self._grads.gW
        self.visible_bias_grad = 0
        self.hidden_bias_grad = self._grads.gd
        self.class_bias_grad = self._grads.gc
        self.class_weights_grad = self._grads.gU

        self.acc_per_epoch_list = []
        self.auc_per_epoch_list = []

        self.data_parallel = _dist_ready() if data_parallel is None else bool(data_parallel)
        self._draw_offset = 0     # Philox draw id, advanced by every sampling call
        self._rows_seen = 0       # global row counter (row0 of the next minibatch)
        self._chain_v = None      # persistent chains (PCD)
        self._chain_y = None
        self._rank_rows = None    # nominal minibatch rows per rank (set by train_rbm from the loader's batch_size)
        self._pset = None
        self._loss_sums = None    # float64 [2] on the device: the running loss sums of train_rbm's current epoch
        self._epoch_graph = None  # (key, hipGraph of one epoch from a device-resident dataset)
        self._draw_base = None    # int64 [2] on the device: base of the draw ids / Philox rows of a captured generative epoch
        if self.data_parallel and hasattr(self.ops, "status_agree"):
            self.ops.status_agree = self._agree_chain_status

    # ------------------------------------------------------------------------------------------
    def _params(self) -> ParamSet:
        # user code may rebind the public attributes, as with the reference: the ParamSet (and its ctypes struct) is kept as
        # long as the five attributes are the same tensor objects
        tensors = (self.weights, self.visible_bias, self.hidden_bias, self.class_weights, self.class_bias)
        if self._pset is None or not self._pset.same_tensors(*tensors):
            self._pset = ParamSet(*tensors)
        return self._pset

    def _next_offset(self, n=1):
        o = self._draw_offset
        self._draw_offset += n
        return o

    # ---- reference API: conditionals (return probabilities, like the reference) ----------------
    def sample_hidden(self, visible_activations, class_activations):
        """src/ClassificationRBM.py:43-47"""
        p, _ = self.ops.sample_hidden(self._params(), visible_activations, class_activations)
        return p

    def sample_visible(self, hidden_activations):
        """src/ClassificationRBM.py:49-52"""
        p, _ = self.ops.sample_visible(self._params(), hidden_activations)
        return p

    def sample_class(self, hidden_activations):
        """src/ClassificationRBM.py:54-60"""
        p, _ = self.ops.sample_class(self._params(), hidden_activations)
        return p

    def sample_class_given_x(self, input_data):
        """src/ClassificationRBM.py:62-86 (p(y|x) from free energies)"""
        _, p = self.ops.free_energy(self._params(), input_data)
        return p

    def free_energy(self, input_data):
        """-F(x, y) for every class: the intermediate of src/ClassificationRBM.py:68-74."""
        nf, _ = self.ops.free_energy(self._params(), input_data)
        return nf

    # ---- Bernoulli / categorical draws (build-defined) ------------------------
    def draw_hidden(self, v, y, row0=0):
        return self.ops.sample_hidden(self._params(), v, y, self.seed, self._next_offset(), row0, True, True)

    def draw_visible(self, h, row0=0):
        return self.ops.sample_visible(self._params(), h, self.seed, self._next_offset(), row0, True, True)

    def draw_class(self, h, row0=0):
        return self.ops.sample_class(self._params(), h, self.seed, self._next_offset(), row0, True, True)

    # ---- reference API: update + training rules ------------------------------------------------
    def update_weights(self, batch_size, factor=1):
        """src/ClassificationRBM.py:88-99, applied to the *_grad buffers."""
        with_gb = not (isinstance(self.visible_bias_grad, (int, float)) and self.visible_bias_grad == 0)
        self.ops.update(self._params(), self._grads, float(self.learning_rate), float(factor), float(batch_size),
                        float(self.sparse_constant), with_gb=with_gb)

    def _allreduce_grads(self):
        """ONE collective per step: the flat gradient buffer, whose last element carries the local row count -- written by
        the gradient launch itself (qbm_rbm_grads.n_rows; a rank with an empty shard contributes zeros: the C entry points
        zero the blocks at B == 0).  The sum stays ON THE DEVICE: qbm_rbm_update_dev reads the global batch size from
        there, so a data-parallel step has no host round trip."""
        import torch.distributed as dist

        dist.all_reduce(self._grads.flat, op=dist.ReduceOp.SUM)  # RCCL over xGMI

    def _agree_chain_status(self, st, off_lo):
        """Collective behind HipRbmOps.check_chain_status: the group's status (max over ranks) and the draw offset of the
        FIRST step any rank's one-launch chain lost (that rank's update was a no-op, and through the -inf row count so was
        every other rank's): all ranks repeat from there."""
        import torch.distributed as dist

        dev = self.device if dist.get_backend() != "gloo" else torch.device("cpu")
        t = torch.tensor([st, (1 << 32) - off_lo if st else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        st, inv = (int(x) for x in t.tolist())
        return st, ((1 << 32) - inv) & 0xFFFFFFFF if st else 0

    def _world(self):
        if not self.data_parallel:
            return 1
        import torch.distributed as dist

        return dist.get_world_size()

    def _rank_row_offset(self, local_rows):
        """Global index, inside the current global minibatch, of this rank's first row: rank * NOMINAL rows per rank
        (the loader's batch_size), so that a short last shard does not reuse the Philox rows of the ranks below it."""
        if not self.data_parallel:
            return 0
        import torch.distributed as dist

        return dist.get_rank() * int(self._rank_rows if self._rank_rows else local_rows)

    def _disc_step(self, input_data, class_label, factor, loss_acc=None, p_out=None, guard_chain=False):
        """One discriminative pass + update (single GPU: everything in three launches; data parallel: + the all-reduce and
        the update launch) -> p(y|x).  loss_acc: float64 [2] device tensor the step adds its two loss terms to.
        guard_chain (hybrid training): the step reads the fused chain launches' status word and is recorded with the pending CD
        steps -- behind a CD step that gave up it stores nothing and HipRbmOps.check_chain_status repeats it in order."""
        batch_size = input_data.size(0)
        P = self._params()
        self.visible_bias_grad = 0  # :138
        lr, sp = float(self.learning_rate), float(self.sparse_constant)
        G, f, bs = self._grads, float(factor), float(batch_size)
        # (`step` must not refer to itself: a function that closes over itself is a reference cycle, and everything it holds --
        # device tensors, the model with its captured hipGraph -- then lives until the cycle collector runs, possibly in the
        # middle of another stream capture)
        if self.data_parallel:
            def step(guard):
                p = self.ops.disc_step(P, G, input_data, class_label, False, False, lr, f, bs, sp, with_count=True, loss_acc=loss_acc,
                                       p_out=p_out, **(dict(guard_chain=True, replay=functools.partial(step, False)) if guard else {}))
                self._allreduce_grads()
                self.ops.update(P, G, lr, f, 0.0, sp, with_gb=False, count_dev=G.count)
                return p
        else:
            def step(guard):
                return self.ops.disc_step(P, G, input_data, class_label, True, not self.keep_grads, lr, f, bs, sp, loss_acc=loss_acc,
                                          p_out=p_out, **(dict(guard_chain=True, replay=functools.partial(step, False)) if guard else {}))
        return step(guard_chain)

    def discriminative_training(self, input_data, class_label, factor=1):
        """src/ClassificationRBM.py:101-146 -> (error, predicted, class_probabilities)"""
        batch_size = input_data.size(0)
        p = self._disc_step(input_data, class_label, factor)
        labels_dev = torch.as_tensor(class_label).to(self.device).long()
        if batch_size == 0:   # an empty shard of a short last minibatch (data-parallel ranks only)
            return torch.zeros((), device=self.device), labels_dev, p
        error = self.loss(p, labels_dev)  # on probabilities, as the reference does (:142)
        _, predicted = torch.max(p, 1)
        return error, predicted, p

    def contrastive_divergence_training(self, input_data, class_label, factor=1, *, onehot=None, loss_acc=None, _rel=None):
        """CD-k (or PCD when persistent=True) generative step, SURVEY.md 8.A-7:
        dW = v0^T p(h0) - vk^T p(hk), db = sum(v0 - vk), dd = sum(p(h0) - p(hk)), dU = y0^T p(h0) - yk^T p(hk),
        dc = sum(y0 - yk); update rule of src/ClassificationRBM.py:88-99.
        onehot: the labels as fp32 one-hot rows if the caller has them (DeviceDataset delivers them with the gather);
        loss_acc: float64 [2] device tensor -- the step then also evaluates p(y|x) of its minibatch under the parameters
        BEFORE its update (what discriminative_training reports) and adds the two loss terms of train_rbm's bookkeeping.
        _rel (train_rbm's captured epochs): (step index in the epoch, [2] device words) -- draw ids and Philox rows are passed
        RELATIVE to the epoch and the device words supply the epoch's base; the host counters are advanced by the caller."""
        batch_size = input_data.size(0)
        P = self._params()
        if onehot is None:
            labels_dev = torch.as_tensor(class_label).to(self.device).long()
            onehot = torch.nn.functional.one_hot(labels_dev, num_classes=self.num_classes).float()
        y0 = onehot
        monitor = None
        if loss_acc is not None:
            lab32 = torch.as_tensor(class_label)
            if lab32.dtype != torch.int32 or lab32.device != self.device:
                lab32 = lab32.to(self.device, torch.int32)
            monitor = (lab32.contiguous(), loss_acc)
        v0 = torch.as_tensor(input_data)
        # uint8 {0,1} minibatches (DeviceDataset with binary storage) stay bytes: the Gibbs kernels read them as such
        v0 = v0.to(self.device) if v0.dtype == torch.uint8 else v0.to(self.device, torch.float32)
        lr, sp = float(self.learning_rate), float(self.sparse_constant)
        nominal = int(self._rank_rows if self._rank_rows else batch_size)
        if _rel is None:
            row0 = (self._rows_seen + self._rank_row_offset(batch_size)) & 0xFFFFFFFF
            offset = self._next_offset(3 * self.k + 1)
            draw_base = None
        else:
            row0 = (_rel[0] * nominal * self._world() + self._rank_row_offset(batch_size)) & 0xFFFFFFFF
            offset = _rel[0] * (3 * self.k + 1)
            draw_base = _rel[1]
        if self.persistent and (self._chain_v is None or self._chain_v.shape[0] < batch_size):
            # chains start at the data.  Chain states are binary (uint8): {0,1} data are taken as they are,
            # greyscale data are thresholded at 0.5 (data_loader.binarise's rule)
            self._chain_v = v0.clone() if v0.dtype == torch.uint8 else (v0 > 0.5).to(torch.uint8)
            self._chain_y = y0.clone()
        # a short (last) minibatch advances the first `batch_size` chains and leaves the others where they are, so the set
        # of chains persists over epochs -- and is the same set however the minibatch is sharded over ranks
        cv = self._chain_v[:batch_size] if self.persistent else None
        cy = self._chain_y[:batch_size] if self.persistent else None
        self.visible_bias_grad = self._grads.gb
        G, k, seed, f, bs = self._grads, self.k, self.seed, float(factor), float(batch_size)
        # `step` is also what HipRbmOps.check_chain_status calls to repeat this minibatch on the launch-per-phase path should
        # its one-launch chain have given up (its gradient and update launches were then no-ops): handed over as
        # functools.partial(step, None) -- a repeat records nothing itself
        if self.data_parallel:
            def step(replay):
                self.ops.cd_step(P, G, v0, y0, k, seed, offset, row0, cv, cy, False, lr, f, bs, sp, with_count=True, replay=replay,
                                 monitor=monitor, draw_base=draw_base)
                self._allreduce_grads()
                self.ops.update(P, G, lr, f, 0.0, sp, count_dev=G.count)   # global batch size read on the device
        elif self.keep_grads:
            def step(replay):
                self.ops.cd_step(P, G, v0, y0, k, seed, offset, row0, cv, cy, False, lr, f, bs, sp, replay=replay, monitor=monitor)
                self.ops.update(P, G, lr, f, bs, sp, guard_chain=True)
        else:  # every update fused into the gradient launch
            def step(replay):
                self.ops.cd_step(P, G, v0, y0, k, seed, offset, row0, cv, cy, True, lr, f, bs, sp, replay=replay, monitor=monitor,
                                 draw_base=draw_base)
        step(functools.partial(step, None))
        # Philox rows advance by the NOMINAL global minibatch (the loader's batch_size x ranks; a short last minibatch
        # included): known on the host of every rank without asking the device for the all-reduced row count
        if _rel is None:
            self._rows_seen += nominal * self._world()
        return None

    def _sigmoid(self, x):
        """src/ClassificationRBM.py:148-149 (host helper kept for API parity)"""
        return 1 / (1 + torch.exp(-x))

    def _random_probabilities(self, num):
        """src/ClassificationRBM.py:151-157: uniforms in [0,1); here from the device Philox stream."""
        return self.ops.uniform(self.seed, self._next_offset(), 0, 1, int(num)).view(-1)

    # ---- reference API: loops ------------------------------------------------------------------
    def train_rbm(self, train_loader, epochs, cuda=False, validation_loader=None, test_loader=None,
                  method='discriminative', generative_factor=None, discriminative_factor=1):
        """src/ClassificationRBM.py:159-205 -> (loss_list, model, nll_list)"""
        print('Training RBM...')
        loss_list = []
        nll_list = []
        best_validation_model = None
        self._rank_rows = getattr(train_loader, 'batch_size', None)   # nominal rows per rank (Philox row offsets)
        if method not in ('discriminative', 'generative', 'hybrid'):
            raise NotImplementedError(method)
        gen_factor = generative_factor if generative_factor is not None else (1 if method == 'generative' else 0.01)
        # a DeviceDataset hands out device tensors as the kernels take them (uint8 / fp32 rows, int32 labels, one-hot rows
        # from the same gather launch): no conversion launches per step
        resident = hasattr(train_loader, 'batches') and hasattr(train_loader, 'epoch')
        if self._loss_sums is None:   # (one pair for the life of the object: a captured epoch graph holds its address)
            self._loss_sums = torch.zeros(2, dtype=torch.float64, device=self.device)
        sums = self._loss_sums
        # Discriminative epochs from a device-resident dataset are captured in ONE hipGraph and replayed: the launches of an
        # epoch have the same arguments in every epoch (the dataset keeps its epoch buffers, the rule draws no random numbers,
        # the loss sums are device memory), so from the second epoch on an epoch costs the host one gather and one replay --
        # the loop then runs at the kernels' speed whatever the host's (eager launches sit at the edge of being host-bound:
        # ~25 us of Python + launch calls per 25 us step).  Same launches, same order: same bits.
        # Data parallel (one process per GPU): the step's RCCL all-reduce of the flat gradient buffer and the update launch that
        # reads the all-reduced row count on the device are captured with it -- every rank captures the same number of steps (a
        # rank whose shard of a short last minibatch is empty still zeroes its gradient blocks and takes part), so the
        # data-parallel loop is not host-bound either (eager: five launches and a collective per step from Python).
        graph_ok = (resident and method in ('discriminative', 'generative') and self.device.type == 'cuda'
                    and getattr(train_loader, 'epoch_gather', False) and hasattr(self.ops, 'lib')
                    and os.environ.get('QBM_EPOCH_GRAPH', '1') != '0')
        # The generative rules draw random numbers: their captured launches carry draw ids and Philox rows RELATIVE to the epoch,
        # and two device words (qbm_rbm_cd_workspace.draw_base) carry the epoch's base -- moved between replays, so the draws
        # are those of the eager loop.  Only where the step is one launch per phase (CD-1 by default: the one-launch chain keeps
        # its fail-safe record on the host, per step) and the update rides in the gradient launch (or, data parallel, behind the
        # collective).
        if graph_ok and method == 'generative':
            from ._lib import QBM_U8
            graph_ok = (self.data_parallel or not self.keep_grads) and not self.ops._may_fuse(self.k, QBM_U8)
        hybrid_guard = False
        if method == 'hybrid' and hasattr(self.ops, '_may_fuse'):
            from ._lib import QBM_U8
            hybrid_guard = bool(self.ops._may_fuse(self.k, QBM_U8))
        epoch_graph = self._epoch_graph   # (kept across calls: a later train_rbm on the same data replays from its first epoch)
        for epoch in range(epochs):
            # The two running sums of the epoch (:187 `epoch_error += batch_error.item()`, :191 `epoch_nll += nll.item()`)
            # live ON THE DEVICE as float64: the gradient launch of every step adds its minibatch's terms (per row
            # CrossEntropyLoss on the probabilities, :142, and -log(p_y + 1e-8), :188-190; fp32 batch means, then the same
            # IEEE double additions in the same order as the reference's Python floats) -- no launch, no host sync per step.
            sums.zero_()
            if resident:
                batches = train_loader.batches(want_onehot=method != 'discriminative')
            else:
                batches = ((b, l, None) for b, l in train_loader)
            if graph_ok and (epoch >= 1 or epoch_graph is not None):
                # (The first epoch of a fresh object runs eagerly: every workspace exists afterwards.)  The generator's first
                # step performs the epoch's gather into the dataset's buffers; the minibatches it yields are the same views in
                # every epoch.
                views = list(batches)
                # everything a captured launch has baked into its arguments (the seed, the gradient buffer and
                # the persistent chains are kernel arguments too)
                key = (id(getattr(train_loader, 'last_views', None)), len(views), views[0][0].data_ptr() if views else 0,
                       views[0][1].data_ptr() if views else 0, getattr(train_loader, 'N', None), getattr(train_loader, 'batch_size', None),
                       views[-1][0].shape[0] if views else 0) + (
                    id(self._params()), float(self.learning_rate), float(self.sparse_constant), float(discriminative_factor), self.keep_grads,
                    method, float(gen_factor), self.k, self.persistent, int(self.seed), id(self._grads), self.data_parallel,
                    0 if self._chain_v is None else self._chain_v.data_ptr(), 0 if self._chain_y is None else self._chain_y.data_ptr())
                if epoch_graph is not None and epoch_graph[0] != key and epoch == 0:
                    # other launches than the captured ones (another model state, data set or rule since the last call): this
                    # epoch runs eagerly -- every workspace of the new shapes exists afterwards, nothing is allocated inside a
                    # capture -- and the next one captures
                    epoch_graph = self._epoch_graph = None
                    batches = iter(views)
                    views = None
                if views is not None and (epoch_graph is None or epoch_graph[0] != key):
                    if self._draw_base is None:
                        self._draw_base = torch.zeros(2, dtype=torch.int64, device=self.device)
                    g = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    # No cycle collection while the stream is capturing: the collector may run at any allocation, and what it
                    # finds may be an older model's hipGraph or device tensors, whose destruction is not a legal call during a
                    # capture (torch.cuda.graph no longer collects by itself) -- found as an abort of the GPU test process in
                    # round 4.  Collect first, then keep the collector off until the capture has ended.
                    gc.collect()
                    gc_was_on = gc.isenabled()
                    gc.disable()
                    try:
                        with torch.cuda.graph(g):
                            for i, (batch, labels, onehot) in enumerate(views):
                                if method == 'discriminative':
                                    self._disc_step(batch, labels, discriminative_factor, sums, False)
                                else:
                                    self.contrastive_divergence_training(batch, labels, gen_factor, onehot=onehot, loss_acc=sums,
                                                                         _rel=(i, self._draw_base))
                    finally:
                        if gc_was_on:
                            gc.enable()
                    # (the graph's launches hold the addresses of the step's scratch: the workspaces of the moment are kept alive with
                    # it, whatever the ops object's caches evict later)
                    keep = (dict(getattr(self.ops, '_cd_wss', {})), dict(getattr(self.ops, '_disc_wss', {})), views, sums, self._draw_base)
                    epoch_graph = self._epoch_graph = (key, g, keep)
                if views is not None:
                    if method == 'generative':   # this epoch's base of the draw ids / Philox rows, then the host counters move on
                        self._draw_base.copy_(torch.tensor([self._draw_offset, self._rows_seen & 0xFFFFFFFF], dtype=torch.int64))
                        self._draw_offset += len(views) * (3 * self.k + 1)
                        self._rows_seen += len(views) * int(self._rank_rows if self._rank_rows else views[0][0].shape[0]) * self._world()
                    epoch_graph[1].replay()
                    batches = ()
            for batch, labels, onehot in batches:
                if not resident:
                    batch = batch.view(len(batch), self.num_visible)  # flatten input data (:173)
                    # uint8 {0,1} minibatches stay bytes: the kernels read them as such
                    batch = batch.to(self.device) if batch.dtype == torch.uint8 else batch.to(self.device, torch.float32)
                    labels = labels.to(self.device, torch.int32)
                if method == 'discriminative':
                    self._disc_step(batch, labels, discriminative_factor, sums, False)
                elif method == 'generative':
                    # p(y|x) is monitored under the parameters BEFORE the step's update, like the discriminative rule's report
                    self.contrastive_divergence_training(batch, labels, gen_factor, onehot=onehot, loss_acc=sums)
                else:
                    # Larochelle & Bengio hybrid objective: discriminative step + generative_factor * CD-k step.  Where the CD
                    # steps may go to a launch that can give up, the discriminative steps between them read the same status
                    # word and are repeated in order with them
                    self._disc_step(batch, labels, discriminative_factor, sums, False, guard_chain=hybrid_guard)
                    self.contrastive_divergence_training(batch, labels, gen_factor, onehot=onehot)
            if resident:
                train_loader.epoch += 1
            if hasattr(self.ops, "check_chain_status"):
                self.ops.check_chain_status()   # the one-launch Gibbs chain: a launch that gave up is repeated here (never silently lost)
            epoch_error, epoch_nll = sums.tolist()   # the epoch's one synchronisation
            loss_list.append(epoch_error / len(train_loader))
            nll_list.append(epoch_nll / len(train_loader))

            if test_loader is not None:
                acc, auc = self.run_test_set(test_loader)
                self.acc_per_epoch_list.append(acc)
                self.auc_per_epoch_list.append(auc)
            print(f'Epoch {epoch} | Error: {epoch_error / len(train_loader):.4f} | NLL: {epoch_nll / len(train_loader):.4f}')
            best_validation_model = self

        return loss_list, best_validation_model, nll_list

    def test_rbm_model(self, rbm_model, test_loader, args):
        """src/ClassificationRBM.py:207-228"""
        correct = 0
        total = 0
        for batch, labels in test_loader:
            batch = batch.view(len(batch), args.visible_units)
            predicted_probabilities = rbm_model.sample_class_given_x(batch)
            _, predicted = torch.max(predicted_probabilities, 1)
            total += labels.size(0)
            correct += (predicted.cpu() == labels.cpu()).sum().item()
        print('Accuracy of the network on the test images: %d %%' % (100 * correct / total))
        return correct / total

    def get_device(self, use_gpu: bool = True) -> torch.device:
        """src/ClassificationRBM.py:230-249 returns CPU unconditionally; tensors handed to this class may
        stay on the CPU (they are staged to the GPU per call), so the reference driver's
        `.to(rbm.get_device(USE_GPU))` keeps working.  The model itself lives on `self.device`."""
        return torch.device("cpu") if not use_gpu else self.device

    def run_test_set(self, test_loader):
        """src/ClassificationRBM.py:251-294 -> (accuracy, roc_auc on hard predictions)"""
        from sklearn.metrics import roc_auc_score

        correct = 0
        total = 0
        all_preds = []
        all_labels = []
        for batch, labels in test_loader:
            batch = batch.view(len(batch), self.num_visible)
            predicted = self._predict_rows(batch).cpu()
            all_preds.extend(predicted.numpy())
            all_labels.extend(labels.cpu().numpy())
            total += labels.size(0)
            correct += (predicted == labels.cpu()).sum().item()
        accuracy = correct / total
        print('Accuracy of the network on the test images: ', accuracy)
        auc_score = roc_auc_score(all_labels, all_preds)
        print('AUC Score: %.4f' % auc_score)
        return accuracy, auc_score

    def _predict_rows(self, batch):
        """argmax_y p(y|x) for every row.  Data parallel: every rank is handed the same evaluation
        batch, classifies only its contiguous share of the rows, and one all-reduce of the [rows] prediction vector
        (zeros outside the own share) gives every rank all predictions -- from which (correct, total) and the AUC
        follow exactly as in the single-process loop."""
        n = batch.size(0)
        if not self.data_parallel or n == 0:
            return torch.max(self.sample_class_given_x(batch), 1)[1]
        import torch.distributed as dist

        rank, world = dist.get_rank(), dist.get_world_size()
        per = (n + world - 1) // world
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        pred = torch.zeros(n, dtype=torch.int64, device=self.device)
        if hi > lo:
            pred[lo:hi] = torch.max(self.sample_class_given_x(batch[lo:hi]), 1)[1]
        if dist.get_backend() == "gloo":   # CPU test harness: the collective runs on host memory
            host = pred.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            return host.to(self.device)
        dist.all_reduce(pred, op=dist.ReduceOp.SUM)
        return pred

    def get_num_params(self):
        """src/ClassificationRBM.py:297-307"""
        return (self.weights.numel() + self.visible_bias.numel() + self.hidden_bias.numel()
                + self.class_weights.numel() + self.class_bias.numel())
