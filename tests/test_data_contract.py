"""CPU: the data_loader mirror reproduces the reference loaders' outputs (shapes, dtypes, order, checksums;
golden values written by tools/make_golden.py from the imported reference, SURVEY.md 8.B boundary B4)."""
import os

import numpy as np

from helpers import GOLDEN, golden


def test_medmnist_contract():
    from qbm_image_classification_amd import data_loader as dl

    c = golden("data_contract.npz")
    (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(os.path.join(GOLDEN, "data_breastmnist.npz"))
    assert trX.dtype == np.float32 and trX.shape[1:] == (28, 28) and trY.dtype == np.uint8
    tr, te, va = dl.preprocess_images(trX, teX, vaX)              # note the (train, test, val) order
    assert tuple(tr.shape) == tuple(c["train_shape"]) and tuple(te.shape) == tuple(c["test_shape"]) \
        and tuple(va.shape) == tuple(c["val_shape"])
    assert str(tr.dtype) == str(c["train_dtype"]) and str(trY.dtype) == str(c["label_dtype"])
    assert np.isclose(tr.astype(np.float64).sum(), float(c["train_sum"])) and np.array_equal(tr[0], c["train_row0"])
    assert np.array_equal(trY, c["train_labels"])
    xs, ys = dl.shuffle_images(tr[:10], trY[:10], seed=44)
    assert np.array_equal(ys, c["shuffle44_labels"]) and np.allclose(xs.sum(1), c["shuffle44_rowsum"])
    assert np.array_equal(dl.encode_labels_to_onehot(np.array([0, 1, 1]), 2), c["onehot"])


def test_mnist_fixture_and_packing():
    from qbm_image_classification_amd import data_loader as dl

    c = golden("data_contract.npz")
    f = golden("data_mnist01_t10k_bits.npz")
    x = dl.unpack_bits(f["bits"], 784)
    assert x.shape == (int(c["mnist01_shape"][0]), 784) and set(np.unique(x)) <= {0, 1}
    assert np.array_equal(dl.pack_bits(x), f["bits"]) and f["bits"].shape[1] == 98
    assert np.array_equal(dl.binarise(np.array([[0.2, 0.5, 0.51]])), [[0, 0, 1]])


def test_bars_and_stripes():
    from qbm_image_classification_amd import data_loader as dl

    x, y = dl.bars_and_stripes(4)
    assert x.shape == (28, 16) and y.sum() == 14 and len({tuple(r) for r in x}) == 28
    for xi, yi in zip(x, y):
        g = xi.reshape(4, 4)
        assert (g == g[0]).all() if yi == 0 else (g == g[:, :1]).all()   # bars: equal rows; stripes: equal columns
