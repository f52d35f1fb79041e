"""GPU: the drop-in classes end to end on the HIP path, against the imported reference's golden run."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_train_rbm_reproduces_reference_run_on_gpu():
    """rbm_main.py's flow (BreastMNIST, H = 9, batch 7, discriminative rule) for 2 epochs: loss, NLL, accuracy, AUC
    and the trained parameters of the reference's CPU run (tools/make_golden.py) are reproduced by the HIP path."""
    from qbm_image_classification_amd import data_loader as dl
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    g = golden("rbm_train_breast.npz")
    (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(os.path.join(GOLDEN, "data_breastmnist.npz"))
    tr, te, va = dl.preprocess_images(trX, teX, vaX)
    n_tr, n_te, seed = int(g["n_tr"]), int(g["n_te"]), int(g["seed"])
    X, Y = tr[:n_tr].astype(np.float32), np.array(trY[:n_tr]).astype(np.int64)
    Xt, Yt = te[:n_te].astype(np.float32), np.array(teY[:n_te]).astype(np.int64)
    for keep in (True, False):   # materialised gradients / update fused into the gradient launch
        np.random.seed(seed); torch.manual_seed(seed)
        r = ClassificationRBM(784, int(g["H"]), 1, num_classes=2, learning_rate=float(g["lr"]), seed=seed, keep_grads=keep)
        assert r.weights.is_cuda and np.array_equal(r.weights.cpu().numpy(), g["W0"])
        tl = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(X), torch.from_numpy(Y)),
                                         batch_size=int(g["batch"]), shuffle=False)
        tel = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(Xt), torch.from_numpy(Yt)),
                                          batch_size=n_tr, shuffle=False)
        loss_list, model, nll_list = r.train_rbm(train_loader=tl, test_loader=tel, epochs=2, method="discriminative")
        assert np.allclose(loss_list, g["loss"], rtol=0, atol=2e-5) and np.allclose(nll_list, g["nll"], rtol=0, atol=2e-5)
        assert np.array_equal(r.acc_per_epoch_list, g["acc"]) and np.array_equal(r.auc_per_epoch_list, g["auc"])
        for got, want in ((r.weights, g["W_after"]), (r.class_weights, g["U_after"]), (r.hidden_bias, g["d_after"]),
                          (r.class_bias, g["c_after"])):
            assert np.allclose(got.cpu().numpy(), want, rtol=0, atol=2e-6)
        p = r.sample_class_given_x(torch.from_numpy(Xt)).cpu().numpy()
        assert np.array_equal(p.argmax(1), g["pyx_test"].argmax(1))        # identical argmax labels
        nf = r.free_energy(torch.from_numpy(Xt)).cpu().numpy()
        assert np.isfinite(nf).all()


def test_reference_api_surface_on_gpu():
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    r = ClassificationRBM(16, 8, 2, num_classes=3, learning_rate=0.1, seed=1)
    v = (torch.rand(5, 16) > 0.5).float()
    y = torch.nn.functional.one_hot(torch.tensor([0, 1, 2, 1, 0]), 3).float()
    ph = r.sample_hidden(v, y); pv = r.sample_visible((torch.rand(5, 8) > 0.5).float()); py = r.sample_class(ph)
    assert ph.shape == (5, 8) and pv.shape == (5, 16) and py.shape == (5, 3)
    assert ((ph >= 0) & (ph <= 1)).all() and torch.allclose(py.sum(1), torch.ones(5, device=py.device), atol=1e-6)
    pyx = r.sample_class_given_x(v)
    assert torch.allclose(pyx.sum(1), torch.ones(5, device=pyx.device), atol=1e-5)
    err, pred, p = r.discriminative_training(v, torch.tensor([0, 1, 2, 1, 0]))
    assert err.ndim == 0 and pred.shape == (5,) and p.shape == (5, 3)
    r.update_weights(5)                                       # re-applies the stored gradients, like the reference API
    p_h, s_h = r.draw_hidden(v, y)
    assert set(torch.unique(s_h).tolist()) <= {0.0, 1.0}
    u = r._random_probabilities(100)
    assert u.shape == (100,) and (u >= 0).all() and (u < 1).all()
    assert r.get_num_params() == 16 * 8 + 16 + 8 + 3 * 8 + 3
    assert r.get_device(False) == torch.device("cpu") and r.get_device(True).type == "cuda"


def test_hybrid_training_learns_bars_and_stripes_on_gpu():
    """BASELINE.json configs[0] shape (16 visible, 8 hidden, CD-1) on the GPU path."""
    from qbm_image_classification_amd import data_loader as dl
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    x, y = dl.bars_and_stripes(4)
    ds = torch.utils.data.TensorDataset(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y))
    tl = torch.utils.data.DataLoader(ds, batch_size=7, shuffle=True, generator=torch.Generator().manual_seed(0))
    tel = torch.utils.data.DataLoader(ds, batch_size=28, shuffle=False)
    r = ClassificationRBM(16, 8, 1, num_classes=2, learning_rate=0.5, seed=42)
    r.train_rbm(train_loader=tl, test_loader=tel, epochs=150, method="hybrid", generative_factor=0.1)
    assert r.acc_per_epoch_list[-1] >= 0.95
    rp = ClassificationRBM(16, 8, 3, num_classes=2, learning_rate=0.1, seed=42, persistent=True)
    rp.train_rbm(train_loader=tl, test_loader=tel, epochs=2, method="generative")
    assert rp._chain_v.dtype == torch.uint8 and torch.isfinite(rp.weights).all()


def test_sample_q_protocol_on_gpu():
    """src/qubo/sampler.py:26-33 protocol: Q upper-triangular, minimise s^T Q s -> float32 [num_reads, n]."""
    from qbm_image_classification_amd.sampler import HipGibbsSampler

    rs = np.random.RandomState(4)
    n = 10
    Q = np.triu(rs.uniform(-1, 1, (n, n)))
    s = HipGibbsSampler(num_sweeps=200, seed=11).sample_Q(Q, 64)
    assert s.shape == (64, n) and s.dtype == np.float32 and set(np.unique(s)) <= {0.0, 1.0}
    import itertools

    states = np.array(list(itertools.product([0, 1], repeat=n)), dtype=np.float64)
    E = np.einsum("si,ij,sj->s", states, Q, states)
    Es = np.einsum("si,ij,sj->s", s.astype(np.float64), Q, s.astype(np.float64))
    assert np.isclose(Es.min(), E.min()) and (np.isclose(Es, E.min())).mean() > 0.5    # annealed reads sit in the ground state
    lin_only = np.diag(rs.uniform(-1, 1, 5))
    s2 = HipGibbsSampler(num_sweeps=50, seed=1).sample_Q(lin_only, 8)                   # the reference's linear-only shortcut case
    assert np.array_equal(s2, np.tile((np.diag(lin_only) < 0).astype(np.float32), (8, 1)))


def test_reference_run_configuration_h9_exact_per_epoch():
    """BASELINE.json's "CPU-ref accuracy match" at the reference's own run configuration: rbm_main.py:32-33 seeds,
    :61-76 H = 9, batch 7, lr default, 20 epochs over the FULL BreastMNIST train split, torch DataLoader(shuffle=True)
    (= rbm_main.py --host-loader here).  Golden: the imported reference (tools/make_golden_runs.py), first 3 seeds.
    Bar: the loader visits the samples in the reference's order; test accuracy AND AUC equal for every one of the 20
    epochs; loss / NLL per epoch to 1e-5; final parameters to 1e-4 (measured: ~1e-6); p(y|x) on the test split to 1e-5
    with identical argmax."""
    from helpers import breastmnist_splits, golden
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    g = golden("rbm_run_breast_h9.npz")
    (tr, trY), _, (te, teY) = breastmnist_splits()
    for seed in (int(s) for s in g["seeds"]):
        np.random.seed(seed); torch.manual_seed(seed)
        r = ClassificationRBM(784, int(g["H"]), 1, num_classes=2, learning_rate=float(g["lr"]), seed=seed, device="cuda:0")
        assert r.ops.name == "hip"
        seen = []

        class Logged(torch.utils.data.TensorDataset):
            def __getitem__(self, i):
                seen.append(int(i))
                return super().__getitem__(i)

        ds = Logged(torch.from_numpy(tr), torch.from_numpy(trY).long())
        tl = torch.utils.data.DataLoader(ds, batch_size=int(g["batch"]), shuffle=True)
        tel = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(te), torch.from_numpy(teY).long()),
                                          batch_size=len(ds), shuffle=False)
        loss, _, nll = r.train_rbm(train_loader=tl, test_loader=tel, epochs=int(g["epochs"]), method="discriminative")
        assert np.array_equal(np.array(seen).reshape(int(g["epochs"]), -1), g[f"s{seed}_order"]), seed
        assert np.array_equal(r.acc_per_epoch_list, g[f"s{seed}_acc"]), (seed, r.acc_per_epoch_list, g[f"s{seed}_acc"])
        assert np.array_equal(r.auc_per_epoch_list, g[f"s{seed}_auc"]), seed
        assert np.allclose(loss, g[f"s{seed}_loss"], rtol=0, atol=1e-5) and np.allclose(nll, g[f"s{seed}_nll"], rtol=0, atol=1e-5)
        for k, a in (("W", r.weights), ("U", r.class_weights), ("d", r.hidden_bias), ("c", r.class_bias), ("b", r.visible_bias)):
            assert np.max(np.abs(a.cpu().numpy() - g[f"s{seed}_{k}"])) <= 1e-4, (seed, k)
        p = r.sample_class_given_x(torch.from_numpy(te)).cpu().numpy()
        assert np.array_equal(p.argmax(1), g[f"s{seed}_pyx_test"].argmax(1))
        assert np.max(np.abs(p - g[f"s{seed}_pyx_test"])) <= 1e-5


def test_reference_run_h20_seed_average_within_0p1_percent():
    """north_star: "accuracy within +-0.1 % of the CPU reference ... over the rbm_main.py seed list".  H = 20 (the 16 526-
    parameter configuration of out/BA_data/RBM_results/rbm_performance_summary.csv:2), all 10 seeds, the minibatches
    replayed in the order the imported reference visited them.  Bar: seed-averaged final test accuracy and AUC within 0.001
    absolute of the golden means (per-seed, per-epoch curves are compared too and reported); the published row
    (0.7859 / 0.6610, made by the authors' torch version, i.e. other shuffles) must be within 0.03 as a sanity check."""
    from helpers import ReplayLoader, breastmnist_splits, golden
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM

    g = golden("rbm_run_breast_h20.npz")
    (tr, trY), _, (te, teY) = breastmnist_splits()
    tel = [(torch.from_numpy(te), torch.from_numpy(teY).long())]
    accs, aucs, exact = [], [], 0
    for seed in (int(s) for s in g["seeds"]):
        np.random.seed(seed); torch.manual_seed(seed)
        r = ClassificationRBM(784, int(g["H"]), 1, num_classes=2, learning_rate=float(g["lr"]), seed=seed, device="cuda:0")
        tl = ReplayLoader(tr, trY, g[f"s{seed}_order"], int(g["batch"]))
        r.train_rbm(train_loader=tl, test_loader=tel, epochs=int(g["epochs"]), method="discriminative")
        accs.append(r.acc_per_epoch_list[-1]); aucs.append(r.auc_per_epoch_list[-1])
        exact += int(np.array_equal(r.acc_per_epoch_list, g[f"s{seed}_acc"]) and np.array_equal(r.auc_per_epoch_list, g[f"s{seed}_auc"]))
    print(f"H=20: {exact}/10 seeds reproduce all 20 epochs exactly; mean acc {np.mean(accs):.6f} (golden {float(g['mean_final_acc']):.6f}), "
          f"mean auc {np.mean(aucs):.6f} (golden {float(g['mean_final_auc']):.6f})")
    assert abs(np.mean(accs) - float(g["mean_final_acc"])) <= 1e-3
    assert abs(np.mean(aucs) - float(g["mean_final_auc"])) <= 1e-3
    assert exact >= 9
    assert abs(np.mean(accs) - float(g["published_acc"])) <= 0.03 and abs(np.mean(aucs) - float(g["published_auc"])) <= 0.03


@pytest.mark.parametrize("beta1", [False, True])
def test_config5_pipeline_coreset_mnist_qbm(beta1):
    """BASELINE.json configs[4] as one thing (tools/qbm_pipeline.py): binarised MNIST 0/1 fixture -> hidden-mean embeddings
    (sample_hidden kernel) -> k_center_greedy coreset of 256 images (GPU picks) -> Disc_QBM 784v / 256h trained for 4 epochs
    with the HIP Gibbs sampler (annealed per-image range, and plain Gibbs at beta = 1 as north_star words it) -> accuracy on
    515 held-out digits.  Measured on MI355X: 0.88 / 0.87 (majority class 0.52); bar 0.80.  The sampler replaces dwave-neal
    (absent): parity with the reference's sampler is unpinned, this pins that the replacement LEARNS at the stated size."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    from qbm_pipeline import run

    r = run(hidden=256, budget=256, epochs=4, reads=20, sweeps=20, lr=0.2, batch=32, beta1=beta1, seed=7)
    print(r)
    assert r["coreset"] == 256 and r["test_images"] == 515
    assert r["test_acc"] >= 0.80 and r["coreset_acc"] >= 0.90, r
    assert r["val_acc_per_epoch"][-1] > r["val_acc_per_epoch"][0]          # it learns over the epochs
