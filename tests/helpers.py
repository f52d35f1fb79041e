"""Shared recipes for the parity tests (inputs are regenerated from seeds, outputs come from
tests/golden/*.npz written by tools/make_golden.py from the imported reference)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def synth_params(V, H, C, seed):
    # same recipe as tools/make_golden.py::synth_params (legacy MT19937 streams are version-stable)
    rs = np.random.RandomState(seed)
    W = (rs.standard_normal((V, H)) * 0.1).astype(np.float32)
    b = (rs.standard_normal(V) * 0.2).astype(np.float32)
    d = (rs.standard_normal(H) * 0.2).astype(np.float32)
    U = (rs.standard_normal((C, H)) * 0.3).astype(np.float32)
    c = (rs.standard_normal(C) * 0.1).astype(np.float32)
    return W, b, d, U, c


def synth_inputs(B, V, H, C, seed):
    rs = np.random.RandomState(seed + 1000)
    v_bin = (rs.random_sample((B, V)) < 0.3).astype(np.float32)
    x_grey = rs.random_sample((B, V)).astype(np.float32) * (rs.random_sample((B, V)) < 0.4)
    x_grey = x_grey.astype(np.float32)
    h_bin = (rs.random_sample((B, H)) < 0.5).astype(np.float32)
    labels = rs.randint(0, C, size=B).astype(np.int64)
    y1h = np.eye(C, dtype=np.float32)[labels]
    return v_bin, x_grey, h_bin, labels, y1h


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))


class ReplayLoader:
    """Minibatches in a RECORDED sample order (one row of `order` per epoch), e.g. the order in which the reference's
    DataLoader(shuffle=True) visited the training set in a golden run (tools/make_golden_runs.py)."""

    def __init__(self, X, y, order, batch_size):
        import torch

        self.X, self.y = torch.as_tensor(X), torch.as_tensor(y).long()
        self.order, self.batch_size, self.epoch = np.asarray(order, dtype=np.int64), int(batch_size), 0

    def __len__(self):
        return (self.order.shape[1] + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        import torch

        idx = torch.from_numpy(self.order[self.epoch])
        self.epoch += 1
        for lo in range(0, len(idx), self.batch_size):
            sel = idx[lo:lo + self.batch_size]
            yield self.X[sel], self.y[sel]


def breastmnist_splits():
    """(train, val, test) of the committed BreastMNIST fixture through the data_loader mirror, flattened as
    rbm_main.py:56-59 does it."""
    from qbm_image_classification_amd import data_loader as dl

    (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(os.path.join(GOLDEN, "data_breastmnist.npz"))
    tr, va, te = dl.preprocess_images(trX, vaX, teX)
    return (tr, np.array(trY)), (va, np.array(vaY)), (te, np.array(teY))
