#!/usr/bin/env python
"""tools/accuracy_context.py -- CONTEXT, PARITY UNPINNED: the reference's published PneumoniaMNIST rows with the sampler /
trainer of this build, set side by side.  The reference's numbers come from an annealer that is absent here (dwave-neal, or
a D-Wave QPU); this build samples with heat-bath annealing on the GPU (A-11 stays "parity unpinned"), so
equality is not expected and not claimed -- the rows only show that the drop-in reaches the same accuracy regime at the same
hyper-parameters.

  qbm_h2    Disc_QBM 784v / 2h, batch 20, 20 epochs, lr 0.45294, 100 reads     reference: src/paper_plots/test_result.txt:1-15
            (simulated annealer "BMS"): accuracy 0.8558, AUC 0.8265
  qbm_h10   Disc_QBM 784v / 10h, batch 73, same                                 reference: out/paper_data/QuCUN_Pneumonia_q/
            third_pneu_result.txt:1-15 (Advantage_system4.1 QPU): accuracy 0.8285, AUC 0.7842
  rbm_h14 / rbm_h20   ClassificationRBM, discriminative rule, rbm_main.py's defaults (lr 0.08715, batch 7, 20 epochs) on the
            PneumoniaMNIST splits                                               reference: out/BA_data/RBM_Pneumonia_results/
            rbm_performance_summary_Pneumonia.csv:6 (11 804 parameters = 14 hidden: 0.8354 / 0.7945) and :2 (16 526 = 20
            hidden: 0.7877 / 0.7383), means over its ten seeds

    python tools/accuracy_context.py [--seeds 19 23 29] [--which qbm_h2 qbm_h10 rbm_h14 rbm_h20] [--sweeps 100] [--epochs 20]
One JSON line per run, one summary line per configuration (Disc_QBM: annealed and plain Gibbs at beta = 1)."""
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
import numpy as np  # noqa: E402
import torch  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "data_pneumoniamnist.npz")
REFERENCE = {"qbm_h2": (0.8558, 0.8265, "src/paper_plots/test_result.txt:1-2 (simulated annealer)"),
             "qbm_h10": (0.8285, 0.7842, "out/paper_data/QuCUN_Pneumonia_q/third_pneu_result.txt:1-2 (QPU)"),
             "rbm_h14": (0.8354, 0.7945, "out/BA_data/RBM_Pneumonia_results/rbm_performance_summary_Pneumonia.csv:6 (mean of 10 seeds)"),
             "rbm_h20": (0.7877, 0.7383, "out/BA_data/RBM_Pneumonia_results/rbm_performance_summary_Pneumonia.csv:2 (mean of 10 seeds)")}
QBM = {"qbm_h2": dict(hidden=2, batch=20), "qbm_h10": dict(hidden=10, batch=73)}
RBM_SEEDS = [1967690937, 2286980494, 3620295971, 1662044193, 1825595160, 3054779705, 900327972, 1620954898, 3699850877, 3492574433]


def splits(order):
    from qbm_image_classification_amd import data_loader as dl

    with contextlib.redirect_stdout(io.StringIO()):
        (trX, trY), (vaX, vaY), (teX, teY) = dl.get_medmnist(FIXTURE)
        if order == "qbm":
            tr, te, va = dl.preprocess_images(trX, teX, vaX)          # qbm_main.py:58 order
        else:
            tr, va, te = dl.preprocess_images(trX, vaX, teX)          # rbm_main.py:56-59 order
    return tr, np.asarray(trY).reshape(-1), te, np.asarray(teY).reshape(-1)


def run_qbm(which, seed, beta1, reads=100, sweeps=100, epochs=20, lr=0.4529451796571889, max_train=None):
    from qbm_image_classification_amd import metrics
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    cfg = QBM[which]
    tr, trY, te, teY = splits("qbm")
    if max_train:
        tr, trY = tr[:max_train], trY[:max_train]
    np.random.seed(seed)
    with tempfile.TemporaryDirectory() as tmp:
        q = Disc_QBM(dim_input=784, num_classes=2, epochs=epochs, n_hidden_nodes=cfg["hidden"], seed=seed, solver="HIP", sample_count=reads,
                     anneal_steps=sweeps, beta_eff=1.0, beta_range=(1.0, 1.0) if beta1 else None, speicherort=tmp + "/", param_string="ctx")
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            q.train_model(tr, trY, te, teY, batch_size=cfg["batch"], learning_rate=lr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pred = q.predict_batch(te)
    with contextlib.redirect_stdout(io.StringIO()):
        acc, f1, prec, rec, auc = metrics.get_metrics(teY, pred, ["negative", "positive"])
    return {"config": which, "sampler": "gibbs beta=1" if beta1 else "annealed (per-image range)", "seed": seed, "test_acc": round(float(acc), 4),
            "test_auc": round(float(auc), 4), "train_seconds": round(dt, 2)}


def run_rbm(which, seed, epochs=20, lr=0.08714599435919934, batch=7, max_train=None):
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.device_data import DeviceDataset

    hidden = int(which.split("_h")[1])
    tr, trY, te, teY = splits("rbm")
    if max_train:
        tr, trY = tr[:max_train], trY[:max_train]
    np.random.seed(seed)
    torch.manual_seed(seed)
    rbm = ClassificationRBM(784, hidden, 1, num_classes=2, learning_rate=lr, seed=seed, device="cuda:0")
    loader = DeviceDataset(tr, trY, batch, num_classes=2, storage="f32", seed=seed, device=str(rbm.device))
    test = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.from_numpy(te), torch.from_numpy(teY).long()),
                                       batch_size=len(te), shuffle=False)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        rbm.train_rbm(train_loader=loader, test_loader=test, epochs=epochs, method="discriminative")
    torch.cuda.synchronize()
    return {"config": which, "trainer": "ClassificationRBM.train_rbm, discriminative", "seed": seed, "test_acc": round(float(rbm.acc_per_epoch_list[-1]), 4),
            "test_auc": round(float(rbm.auc_per_epoch_list[-1]), 4), "train_seconds": round(time.perf_counter() - t0, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[19, 23, 29])
    ap.add_argument("--which", nargs="+", default=list(REFERENCE))
    ap.add_argument("--sweeps", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=20)
    a = ap.parse_args()
    yte = splits("qbm")[3]
    majority = round(float(max(yte.mean(), 1 - yte.mean())), 4)
    for which in a.which:
        groups = {}
        if which in QBM:
            for beta1 in (False, True):
                for s in a.seeds:
                    r = run_qbm(which, s, beta1, sweeps=a.sweeps, epochs=a.epochs)
                    groups.setdefault(r["sampler"], []).append(r)
                    print(json.dumps(r), flush=True)
        else:
            for s in RBM_SEEDS[:len(a.seeds)]:
                r = run_rbm(which, s, epochs=a.epochs)
                groups.setdefault(r["trainer"], []).append(r)
                print(json.dumps(r), flush=True)
        ref = REFERENCE[which]
        for name, rs in groups.items():
            print(json.dumps({"summary": which, "variant": name, "mean_test_acc": round(float(np.mean([x["test_acc"] for x in rs])), 4),
                              "mean_test_auc": round(float(np.mean([x["test_auc"] for x in rs])), 4), "runs": len(rs),
                              "reference": {"acc": ref[0], "auc": ref[1], "where": ref[2]}, "majority_class_rate": majority,
                              "status": "context, parity unpinned (the reference's annealer / its per-run permutations are not reproducible here)"}),
                  flush=True)


if __name__ == "__main__":
    main()
