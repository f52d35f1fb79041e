"""CPU: host logic of the ClassificationRBM mirror (constructor draw order, train_rbm / run_test_set loops,
loss and NLL bookkeeping) against golden vectors produced by the imported reference, with an oracle-backed
test double standing in for the HIP ops."""
import os

import numpy as np
import torch

from helpers import GOLDEN, golden
from oracle_backends import OracleRbmOps


def _rbm(*a, **k):
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    return ClassificationRBM(*a, ops=OracleRbmOps(), data_parallel=False, **k)


def test_constructor_state_matches_reference():
    g = golden("rbm_init.npz")
    for seed in (42, 1967690937):
        r = _rbm(16, 8, 1, num_classes=2, seed=seed)
        assert np.array_equal(r.weights.numpy(), g[f"s{seed}_small_W"])          # one torch.randn(V,H) after manual_seed
        assert np.array_equal(r.visible_bias.numpy(), g[f"s{seed}_small_b"])
        assert np.array_equal(r.hidden_bias.numpy(), g[f"s{seed}_small_d"])
        assert np.array_equal(r.class_weights.numpy(), g[f"s{seed}_small_U"])
        assert np.array_equal(r.class_bias.numpy(), g[f"s{seed}_small_c"])
        r = _rbm(784, 128, 1, num_classes=2, seed=seed)
        w = r.weights.numpy()
        assert np.array_equal(w[:4, :8], g[f"s{seed}_mid_W_corner"])
        assert np.allclose([w.astype(np.float64).sum(), (w.astype(np.float64) ** 2).sum()], g[f"s{seed}_mid_W_sum"])
    assert r.get_num_params() == 784 * 128 + 784 + 128 + 2 * 128 + 2


def test_train_rbm_reproduces_reference_run():
    """2 epochs of discriminative training on the 63-image BreastMNIST subset the golden run used."""
    from qbm_image_classification_amd import data_loader as dl

    g = golden("rbm_train_breast.npz")
    (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(os.path.join(GOLDEN, "data_breastmnist.npz"))
    tr, te, va = dl.preprocess_images(trX, teX, vaX)
    n_tr, n_te, seed = int(g["n_tr"]), int(g["n_te"]), int(g["seed"])
    X, Y = tr[:n_tr].astype(np.float32), np.array(trY[:n_tr]).astype(np.int64)
    Xt, Yt = te[:n_te].astype(np.float32), np.array(teY[:n_te]).astype(np.int64)
    np.random.seed(seed); torch.manual_seed(seed)
    r = _rbm(784, int(g["H"]), 1, num_classes=2, learning_rate=float(g["lr"]), seed=seed)
    assert np.array_equal(r.weights.numpy(), g["W0"])
    tl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)),
                                     batch_size=int(g["batch"]), shuffle=False)
    tel = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(Xt), torch.from_numpy(Yt)),
                                      batch_size=n_tr, shuffle=False)
    loss_list, model, nll_list = r.train_rbm(train_loader=tl, test_loader=tel, epochs=2, method="discriminative")
    assert model is r
    assert np.allclose(loss_list, g["loss"], rtol=0, atol=2e-5)      # CrossEntropyLoss on probabilities, as the reference
    assert np.allclose(nll_list, g["nll"], rtol=0, atol=2e-5)
    assert np.array_equal(r.acc_per_epoch_list, g["acc"]) and np.array_equal(r.auc_per_epoch_list, g["auc"])
    for got, want in ((r.weights, g["W_after"]), (r.class_weights, g["U_after"]), (r.hidden_bias, g["d_after"]),
                      (r.class_bias, g["c_after"])):
        assert np.allclose(got.numpy(), want, rtol=0, atol=2e-6)
    p = r.sample_class_given_x(torch.from_numpy(Xt)).numpy()
    assert np.array_equal(p.argmax(1), g["pyx_test"].argmax(1)) and np.allclose(p, g["pyx_test"], rtol=0, atol=2e-4)
    assert r.get_num_params() == int(g["num_params"])


def test_generative_and_hybrid_training_run_and_learn():
    """CD-k / PCD (declared by the reference, rejected at src/ClassificationRBM.py:181-182): on Bars-and-Stripes
    (BASELINE.json configs[0]: 16 visible, 8 hidden, CD-1) the hybrid rule classifies all 28 patterns."""
    from qbm_image_classification_amd import data_loader as dl

    x, y = dl.bars_and_stripes(4)
    ds = torch.utils.data.TensorDataset(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y))
    g = torch.Generator().manual_seed(0)
    tl = torch.utils.data.DataLoader(ds, batch_size=7, shuffle=True, generator=g)
    tel = torch.utils.data.DataLoader(ds, batch_size=28, shuffle=False)
    r = _rbm(16, 8, 1, num_classes=2, learning_rate=0.5, seed=42)
    r.train_rbm(train_loader=tl, test_loader=tel, epochs=150, method="hybrid", generative_factor=0.1)
    assert r.acc_per_epoch_list[-1] >= 0.95
    r2 = _rbm(16, 8, 2, num_classes=2, learning_rate=0.1, seed=42, persistent=True)
    w0 = r2.weights.clone()
    r2.train_rbm(train_loader=tl, test_loader=tel, epochs=2, method="generative")
    assert not torch.equal(w0, r2.weights) and torch.isfinite(r2.weights).all()
    assert r2._chain_v.dtype == torch.uint8 and set(np.unique(r2._chain_v.numpy())) <= {0, 1}
    try:
        r2.train_rbm(train_loader=tl, epochs=1, method="bogus")
        assert False
    except NotImplementedError:
        pass


def test_full_reference_run_reproduced_on_host_logic():
    """The reference's real run configuration (rbm_main.py:32-33, :61-76: full BreastMNIST train split, H = 9, batch 7,
    20 epochs, DataLoader(shuffle=True) after the same manual_seed) for the first seed, golden from the imported
    reference (tools/make_golden_runs.py).  With the oracle standing in for the HIP ops: the torch DataLoader visits
    the samples in the reference's order, test accuracy and AUC agree for EVERY epoch, parameters to 1e-5."""
    from helpers import breastmnist_splits

    g = golden("rbm_run_breast_h9.npz")
    (tr, trY), _, (te, teY) = breastmnist_splits()
    seed = int(g["seeds"][0])
    np.random.seed(seed); torch.manual_seed(seed)
    r = _rbm(784, int(g["H"]), 1, num_classes=2, learning_rate=float(g["lr"]), seed=seed)
    assert np.array_equal(r.weights.numpy()[:4, :4], g[f"s{seed}_W0_corner"])
    seen = []

    class Logged(torch.utils.data.TensorDataset):
        def __getitem__(self, i):
            seen.append(int(i))
            return super().__getitem__(i)

    ds = Logged(torch.from_numpy(tr), torch.from_numpy(trY).long())
    tl = torch.utils.data.DataLoader(ds, batch_size=int(g["batch"]), shuffle=True)          # rbm_main.py --host-loader
    tel = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(te), torch.from_numpy(teY).long()),
                                      batch_size=len(ds), shuffle=False)
    loss, _, nll = r.train_rbm(train_loader=tl, test_loader=tel, epochs=int(g["epochs"]), method="discriminative")
    assert np.array_equal(np.array(seen).reshape(int(g["epochs"]), -1), g[f"s{seed}_order"])
    assert np.array_equal(r.acc_per_epoch_list, g[f"s{seed}_acc"]) and np.array_equal(r.auc_per_epoch_list, g[f"s{seed}_auc"])
    assert np.allclose(loss, g[f"s{seed}_loss"], rtol=0, atol=1e-6) and np.allclose(nll, g[f"s{seed}_nll"], rtol=0, atol=1e-6)
    for k, a in (("W", r.weights), ("U", r.class_weights), ("d", r.hidden_bias), ("c", r.class_bias), ("b", r.visible_bias)):
        assert np.max(np.abs(a.numpy() - g[f"s{seed}_{k}"])) <= 1e-5, k
