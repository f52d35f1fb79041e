"""Device-resident dataset pipeline behind the `data_loader` outputs.

The reference moves the whole training set to the device and lets `torch.utils.data.DataLoader(shuffle=True)`
slice it on the host side (`rbm_main.py:70-74`); every step then pays host-side indexing, and binary data
travel as fp32.  `DeviceDataset` keeps the set in HBM in its smallest exact form (fp32 greyscale, uint8 {0,1},
or bit-packed: 98 bytes per 28x28 image), makes the epoch permutation on the device (stable argsort of
Philox keys: the same permutation for any rank count) and delivers the minibatches of an epoch with ONE HIP gather
launch per epoch (rows + int32 labels + one-hot labels of all of this rank's rows, in epoch order: 47 MB for MNIST as
uint8 -- HBM is 288 GB; the minibatches are then contiguous slices of it and a training step pays no launch for its
data) or, with `epoch_gather=False`, one gather launch per minibatch.  Iterating it yields `(batch, labels)` like the reference's loader,
so `ClassificationRBM.train_rbm(train_loader=DeviceDataset(...))` runs unchanged, with no host copy per step.

There is no CPU fallback: the HIP library must be present (`_lib.load()` raises otherwise).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import QBM_F32, QBM_U8, check, current_stream, ptr

QBM_BITS = 2


class HipDataOps:
    """ctypes binding of the qbm_data_* entry points (include/qbm_hip.h)."""

    def __init__(self, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.QbmHipError("no ROCm device visible: the HIP path cannot run (there is no CPU fallback)")
        self.device = torch.device(device)

    def binarise(self, x, threshold=0.5):
        x = x.to(self.device, torch.float32).contiguous()
        out = torch.empty(x.shape, dtype=torch.uint8, device=self.device)
        check(self.lib.qbm_data_binarise(ptr(x), x.numel(), float(threshold), ptr(out), current_stream()), "qbm_data_binarise")
        return out

    def pack_bits(self, x_u8):
        x_u8 = x_u8.to(self.device, torch.uint8).contiguous()
        n, v = x_u8.shape
        out = torch.empty(n, (v + 7) // 8, dtype=torch.uint8, device=self.device)
        check(self.lib.qbm_data_pack_bits(ptr(x_u8), n, v, ptr(out), current_stream()), "qbm_data_pack_bits")
        return out

    def shuffle_keys(self, seed, epoch, n):
        keys = torch.empty(n, dtype=torch.int32, device=self.device)
        check(self.lib.qbm_data_shuffle_keys(int(seed), int(epoch), n, ptr(keys), current_stream()), "qbm_data_shuffle_keys")
        return keys

    def permutation(self, seed, epoch, n):
        """Epoch permutation: stable ascending argsort of the Philox keys (int32 indices)."""
        return torch.argsort(self.shuffle_keys(seed, epoch, n), stable=True).to(torch.int32)

    def gather(self, src, src_kind, idx, num_visible, dst_kind=QBM_F32, labels=None, num_classes=0, want_onehot=False, out=None):
        """out: (dst, lab, one) buffers of the right shapes to write into (an epoch's buffers, kept by the dataset so that their
        addresses do not change from epoch to epoch), or None = fresh tensors."""
        idx = idx.to(self.device, torch.int32).contiguous()
        b = idx.numel()
        if out is not None:
            dst, lab, one = out
        else:
            dst = torch.empty(b, num_visible, dtype=torch.float32 if dst_kind == QBM_F32 else torch.uint8, device=self.device)
            lab = torch.empty(b, dtype=torch.int32, device=self.device) if labels is not None else None
            one = torch.empty(b, num_classes, dtype=torch.float32, device=self.device) if (labels is not None and want_onehot) else None
        check(self.lib.qbm_data_gather(ptr(src), src_kind, ptr(idx), b, num_visible, ptr(dst), dst_kind, ptr(labels), ptr(lab),
                                       ptr(one), int(num_classes), current_stream()), "qbm_data_gather")
        return dst, lab, one


class DeviceDataset:
    """Training set resident on the GPU; iterate for `(batch, labels)` minibatches.

    x: [N, ...] images (numpy or torch; flattened to [N, V]); y: [N] integer labels.
    storage: "f32" keeps greyscale floats (what the reference trains the RBM on), "u8" keeps {0,1} bytes,
             "bits" keeps 1 bit per unit; "u8"/"bits" binarise with `threshold` unless x is already integer {0,1}.
    out: "f32" or "u8" -- dtype of the delivered minibatch (u8 only for binary storage).
    shuffle/seed: epoch e uses the permutation `HipDataOps.permutation(seed, e, N)`; the epoch counter advances
             with every full iteration.
    epoch_gather: True (default) = one gather launch per epoch into a buffer of this rank's rows, minibatches are slices
             of it; False = one gather launch per minibatch (smallest footprint).  Same values either way.
    rank/world: data parallelism -- every rank walks the same global batches of `batch_size * world` rows and takes
             rows [rank*batch_size, (rank+1)*batch_size) of each, so the union over ranks is the single-process batch.
    """

    def __init__(self, x, y, batch_size, num_classes, storage="f32", out=None, shuffle=True, seed=44, threshold=0.5,
                 drop_last=False, device="cuda:0", rank=0, world=1, ops=None, epoch_gather=True):
        if storage not in ("f32", "u8", "bits"):
            raise ValueError("storage must be 'f32', 'u8' or 'bits'")
        self.ops = ops if ops is not None else HipDataOps(device)
        dev = self.ops.device
        x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
        x = x.reshape(x.shape[0], -1)
        self.N, self.V = int(x.shape[0]), int(x.shape[1])
        self.num_classes = int(num_classes)
        self.batch_size, self.shuffle, self.seed, self.drop_last = int(batch_size), bool(shuffle), int(seed), bool(drop_last)
        self.rank, self.world = int(rank), int(world)
        self.epoch_gather = bool(epoch_gather)
        self.epoch = 0
        self._epoch_bufs = {}
        self.labels = torch.as_tensor(np.asarray(y) if not torch.is_tensor(y) else y).reshape(-1).to(dev, torch.int32).contiguous()
        if self.labels.numel() != self.N:
            raise ValueError("x and y disagree on the number of samples")
        if storage == "f32":
            self.data, self.kind = x.to(dev, torch.float32).contiguous(), QBM_F32
        else:
            xb = x.to(dev)
            xb = xb.to(torch.uint8).contiguous() if not xb.is_floating_point() else self.ops.binarise(xb, threshold)
            if storage == "u8":
                self.data, self.kind = xb, QBM_U8
            else:
                self.data, self.kind = self.ops.pack_bits(xb), QBM_BITS
        self.out_kind = QBM_F32 if (out or ("f32" if storage == "f32" else "u8")) == "f32" else QBM_U8
        if self.out_kind == QBM_U8 and self.kind == QBM_F32:
            raise ValueError("uint8 minibatches need binary storage ('u8' or 'bits')")

    def bytes_resident(self):
        return self.data.numel() * self.data.element_size() + self.labels.numel() * 4

    def __len__(self):
        g = self.batch_size * self.world
        return self.N // g if self.drop_last else (self.N + g - 1) // g

    def order(self, epoch=None):
        """Sample order of an epoch (int32 device tensor)."""
        e = self.epoch if epoch is None else epoch
        if self.shuffle:
            return self.ops.permutation(self.seed, e, self.N)
        return torch.arange(self.N, dtype=torch.int32, device=self.ops.device)

    def batches(self, epoch=None, want_onehot=False):
        """Yield (x[B,V], labels int32 [B], onehot [B,C] or None) for this rank."""
        order = self.order(epoch)
        g = self.batch_size * self.world
        if self.epoch_gather:
            # this rank's rows of every global batch, in epoch order: positions p with (p mod g) in [rank*B, (rank+1)*B)
            n_used = len(self) * g if self.drop_last else self.N
            mine = order[:min(n_used, self.N)]
            if self.world > 1:
                pos = torch.arange(mine.numel(), device=mine.device) % g
                mine = mine[(pos >= self.rank * self.batch_size) & (pos < (self.rank + 1) * self.batch_size)]
            # the epoch's buffers are kept: a minibatch is the SAME slice of the same storage in every epoch (a training loop may
            # capture an epoch's launches in a hipGraph once and replay it: ClassificationRBM.train_rbm)
            nb = int(mine.numel())
            key = (nb, bool(want_onehot))
            if nb and self._epoch_bufs.get("key") != key:
                dt = torch.float32 if self.out_kind == QBM_F32 else torch.uint8
                dev = self.ops.device
                self._epoch_bufs = {"key": key, "bufs": (torch.empty(nb, self.V, dtype=dt, device=dev), torch.empty(nb, dtype=torch.int32, device=dev),
                                                        torch.empty(nb, self.num_classes, dtype=torch.float32, device=dev) if want_onehot else None)}
            X, L, O = (self.ops.gather(self.data, self.kind, mine, self.V, self.out_kind, self.labels, self.num_classes, want_onehot,
                                       out=self._epoch_bufs["bufs"])
                       if mine.numel() else (torch.empty(0, self.V, dtype=torch.float32 if self.out_kind == QBM_F32 else torch.uint8,
                                                         device=self.ops.device), torch.empty(0, dtype=torch.int32, device=self.ops.device), None))
            # the minibatch views of the epoch buffers are made once (a torch slice costs ~2 us: three per step would be a fifth
            # of a training step) and handed out again in every epoch: the same objects, the same storage, new contents
            views = self._epoch_bufs.get("views") if nb else None
            if views is None:
                views, at = [], 0
                for k in range(len(self)):
                    lo = k * g + self.rank * self.batch_size
                    n = max(0, min(lo + self.batch_size, min((k + 1) * g, self.N)) - lo)
                    views.append((X[at:at + n], L[at:at + n], (O[at:at + n] if O is not None else None)))
                    at += n
                if nb:
                    self._epoch_bufs["views"] = views
            self.last_views = views   # (identity = "the same launches as last epoch" for a loop that captures them)
            yield from views
            return
        for k in range(len(self)):
            lo = k * g + self.rank * self.batch_size
            hi = min(lo + self.batch_size, min((k + 1) * g, self.N))
            if hi <= lo:   # a short last global batch may leave the high ranks without rows
                yield (torch.empty(0, self.V, dtype=torch.float32 if self.out_kind == QBM_F32 else torch.uint8, device=self.ops.device),
                       torch.empty(0, dtype=torch.int32, device=self.ops.device), None)
                continue
            yield self.ops.gather(self.data, self.kind, order[lo:hi], self.V, self.out_kind, self.labels, self.num_classes, want_onehot)

    def __iter__(self):
        for xb, lab, _ in self.batches():
            yield xb, lab.long()
        self.epoch += 1
