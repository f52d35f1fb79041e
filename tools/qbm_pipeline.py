#!/usr/bin/env python
"""BASELINE.json configs[4] as ONE pipeline on the GPU: binarised MNIST 0/1 (the committed fixture: the reference checkout
only holds the t10k images) -> hidden-mean embeddings -> k-center-greedy coreset (src/coreset.py:109-120 takes its embeddings
from a light model's last hidden layer: here the hidden means p(h|x) of a ClassificationRBM) -> Disc_QBM 784v / 256h trained
on the coreset with the HIP Gibbs sampler in place of the annealer -> accuracy on held-out digits.

    python tools/qbm_pipeline.py [--hidden 256] [--budget 256] [--epochs 4] [--reads 20] [--sweeps 20] [--lr 0.2]
                                 [--batch 32] [--beta1]      (--beta1: plain Gibbs at beta = 1 instead of annealing)

Prints one JSON line with the accuracies and the time per training minibatch."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def load_mnist01():
    from qbm_image_classification_amd import data_loader

    f = np.load(os.path.join(ROOT, "tests", "golden", "data_mnist01_t10k_bits.npz"))
    x = data_loader.unpack_bits(f["bits"], 784).astype(np.float64)
    return x, f["labels"].astype(np.int64)


def run(hidden=256, budget=256, epochs=4, reads=20, sweeps=20, lr=0.2, batch=32, beta1=False, seed=7, n_train=1600, out_dir=None,
        quiet=True):
    import contextlib
    import io
    import tempfile

    from qbm_image_classification_amd import coreset
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    X, y = load_mnist01()
    Xtr, ytr, Xte, yte = X[:n_train], y[:n_train], X[n_train:], y[n_train:]
    # 1. embeddings: hidden means of a (randomly initialised) RBM -- what the path's sample_hidden kernel produces
    rbm = ClassificationRBM(784, 64, 1, num_classes=2, seed=seed)
    emb = rbm.ops.sample_hidden(rbm._params(), torch.from_numpy(Xtr.astype(np.float32)))[0]
    # 2. coreset (k-center greedy on the GPU)
    idx = coreset.k_center_greedy(emb, budget, random_seed=seed)
    Xc, yc = Xtr[idx], ytr[idx]
    # 3. Disc_QBM on the coreset
    np.random.seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        q = Disc_QBM(dim_input=784, num_classes=2, epochs=epochs, n_hidden_nodes=hidden, seed=seed, solver="HIP", sample_count=reads,
                     anneal_steps=sweeps, beta_eff=1.0, beta_range=(1.0, 1.0) if beta1 else None,
                     speicherort=(out_dir or tmp) + "/", param_string="pipe")
        sink = io.StringIO() if quiet else sys.stdout
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(sink):
            q.train_model(Xc, yc, Xte[:200], yte[:200], batch_size=batch, learning_rate=lr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        acc_test = float((q.predict_batch(Xte) == yte).mean())
        acc_core = float((q.predict_batch(Xc) == yc).mean())
    return {"hidden": hidden, "coreset": int(len(idx)), "coreset_class_balance": float(yc.mean()), "epochs": epochs, "reads": reads,
            "sweeps": sweeps, "lr": lr, "batch": batch, "sampler": "gibbs beta=1" if beta1 else "annealed (per-image range)",
            "val_acc_per_epoch": [round(a, 4) for a in q.training_history.acc_per_epoch], "test_acc": round(acc_test, 4),
            "coreset_acc": round(acc_core, 4), "test_images": int(len(yte)), "train_seconds": round(dt, 2),
            "majority_class_rate": round(float(max(yte.mean(), 1 - yte.mean())), 4)}


def time_iteration(B=73, R=100, sweeps=100, hidden=256, reps=10):
    """train_for_one_iteration at the configuration quoted for BASELINE.json configs[4]: 784v / 256h, 73 images, 100 reads."""
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    rs = np.random.RandomState(0)
    X = (rs.random_sample((B, 784)) < 0.13).astype(np.float64)
    Y = rs.randint(0, 2, B)
    np.random.seed(3)
    q = Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=hidden, seed=5, solver="HIP", sample_count=R, anneal_steps=sweeps)
    Xd, Yd = q._to_dev(X, Y)
    for _ in range(2):
        q._step_device(Xd, Yd, 0.01)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        q._step_device(Xd, Yd, 0.01)
    torch.cuda.synchronize()
    dev_ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        q.train_for_one_iteration(X, Y, 0.01)      # reference API: host arrays in, host scalars out (one sync per call)
    api_ms = (time.perf_counter() - t0) / reps * 1e3
    return {"what": "Disc_QBM.train_for_one_iteration", "B": B, "R": R, "sweeps": sweeps, "hidden": hidden,
            "ms_device_resident_step": round(dev_ms, 3), "ms_reference_api_call": round(api_ms, 3),
            "chain_sweeps_per_s": round(2 * B * R * sweeps / dev_ms * 1e3, 1),
            "spin_updates_per_s": round(B * R * sweeps * (2 * hidden + 1) / dev_ms * 1e3, 1)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    for name, typ, default in (("hidden", int, 256), ("budget", int, 256), ("epochs", int, 4), ("reads", int, 20), ("sweeps", int, 20),
                               ("lr", float, 0.2), ("batch", int, 32), ("seed", int, 7)):
        ap.add_argument("--" + name, type=typ, default=default)
    ap.add_argument("--beta1", action="store_true")
    ap.add_argument("--time", action="store_true", help="also time one training minibatch at B=73, R=100, 100 and 1000 sweeps")
    a = ap.parse_args()
    print(json.dumps(run(a.hidden, a.budget, a.epochs, a.reads, a.sweeps, a.lr, a.batch, a.beta1, a.seed)), flush=True)
    if a.time:
        for sw in (100, 1000):
            print(json.dumps(time_iteration(sweeps=sw)), flush=True)
