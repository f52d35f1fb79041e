#!/usr/bin/env python
"""Context run (not a parity claim): Disc_QBM on BreastMNIST with the hyper-parameters of the reference's published QPU
run (out/BA_data/Breast_beta_hyper_final/1967690937_result.txt there: 5 hidden nodes, batch 36, 20 epochs, learning rate
0.4829; test accuracy 0.7885, ROC AUC 0.6222 with samples from a D-Wave Advantage at beta_eff 27.5), the sampler replaced
by the HIP Gibbs kernels at beta_eff = 1: annealed per-image schedule and plain Gibbs at beta = 1.  The reference's
annealer / QPU are absent (A-11 stays "parity unpinned"), so the numbers can only be set side by side.

    python tools/qbm_breast_context.py [--seeds 19 23 29] [--reads 100] [--sweeps 100]
Prints one JSON line per (sampler, seed) and a summary line."""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def run(seed, beta1, reads, sweeps, hidden=5, batch=36, epochs=20, lr=0.4829083814825144):
    from qbm_image_classification_amd import data_loader as dl, metrics
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    with contextlib.redirect_stdout(io.StringIO()):
        (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(os.path.join(ROOT, "tests", "golden", "data_breastmnist.npz"))
        tr, te, va = dl.preprocess_images(trX, teX, vaX)          # qbm_main.py:58 order
    trY, teY = np.asarray(trY).reshape(-1), np.asarray(teY).reshape(-1)
    np.random.seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        q = Disc_QBM(dim_input=784, num_classes=2, epochs=epochs, n_hidden_nodes=hidden, seed=seed, solver="HIP", sample_count=reads,
                     anneal_steps=sweeps, beta_eff=1.0, beta_range=(1.0, 1.0) if beta1 else None, speicherort=tmp + "/",
                     param_string="ctx")
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            q.train_model(tr, trY, te, teY, batch_size=batch, learning_rate=lr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pred = q.predict_batch(te)
    with contextlib.redirect_stdout(io.StringIO()):
        acc, f1, prec, rec, auc = metrics.get_metrics(teY, pred, ["negative", "positive"])
    return {"sampler": "gibbs beta=1" if beta1 else "annealed (per-image range)", "seed": seed, "test_acc": round(float(acc), 4),
            "test_auc": round(float(auc), 4), "f1": round(float(f1), 4), "train_seconds": round(dt, 2),
            "acc_per_epoch_last3": [round(float(a), 4) for a in q.training_history.acc_per_epoch[-3:]]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[19, 23, 29])
    ap.add_argument("--reads", type=int, default=100)
    ap.add_argument("--sweeps", type=int, default=100)
    a = ap.parse_args()
    res = []
    for beta1 in (False, True):
        for s in a.seeds:
            res.append(run(s, beta1, a.reads, a.sweeps))
            print(json.dumps(res[-1]), flush=True)
    for name in ("annealed (per-image range)", "gibbs beta=1"):
        r = [x for x in res if x["sampler"] == name]
        print(json.dumps({"summary": name, "mean_test_acc": round(float(np.mean([x["test_acc"] for x in r])), 4),
                          "mean_test_auc": round(float(np.mean([x["test_auc"] for x in r])), 4), "seeds": len(r),
                          "reference_qpu_run": {"acc": 0.7885, "auc": 0.6222}, "majority_class_rate": 0.7308}), flush=True)
