#!/usr/bin/env python
"""tools/make_golden_runs.py -- golden END-TO-END runs of the reference's rbm_main.py flow, made by IMPORTING THE REAL
REFERENCE (build container only; needs /root/reference, read-only).  Only numbers are written.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden_runs.py

The flow of /root/reference/rbm_main.py:36-76 is followed statement by statement (seed list :32-33, np/torch seeding
:37-38, get_medmnist + preprocess_images :56-59, ClassificationRBM(784, H, k=1, lr) :61, TensorDataset / DataLoader
(batch 7, shuffle=True) :70-74, train_rbm(20 epochs, 'discriminative') :76); the script itself is not executed because
it parses sys.argv at import and writes into a directory that does not exist.  The order in
which DataLoader(shuffle=True) visits the samples is recorded through a TensorDataset subclass that logs the indices it
is asked for (no extra RNG draw), so that a test can replay the very same minibatches.

  rbm_run_breast_h9.npz   H = 9 (the script's default), the first 3 seeds: per-epoch sample order, loss, NLL, test
                          accuracy / AUC, final parameters
  rbm_run_breast_h20.npz  H = 20 (the configuration of out/BA_data/RBM_results/rbm_performance_summary.csv:2: 16 526
                          parameters), all 10 seeds: per-epoch sample order, test accuracy / AUC, and the published row
"""
import contextlib
import io
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(1)  # deterministic summation order inside the reference's torch ops

SEEDS = [1967690937, 2286980494, 3620295971, 1662044193, 1825595160, 3054779705, 900327972, 1620954898,
         3699850877, 3492574433]                                  # rbm_main.py:32-33
LR, EPOCHS, BATCH = 0.08714599435919934, 20, 7                   # rbm_main.py:14-16


class LoggingTensorDataset(torch.utils.data.TensorDataset):
    """TensorDataset that remembers which indices were fetched, in order."""

    def __init__(self, *tensors):
        super().__init__(*tensors)
        self.log = []

    def __getitem__(self, index):
        self.log.append(int(index))
        return super().__getitem__(index)


def run(seed, hidden):
    from src import data_loader
    from src.ClassificationRBM import ClassificationRBM

    np.random.seed(seed)
    torch.manual_seed(seed)
    (train_X, train_y), (val_X, val_y), (test_X, test_y) = data_loader.get_medmnist("src/data/medmnist/breastmnist.npz")
    train_X, val_X, test_X = data_loader.preprocess_images(train_X, val_X, test_X)
    rbm = ClassificationRBM(784, hidden, 1, num_classes=2, learning_rate=LR, use_cuda=False, seed=seed)
    device = rbm.get_device(False)
    train_y = torch.from_numpy(np.array(train_y)).to(device).long()
    test_y = torch.from_numpy(np.array(test_y)).to(device).long()
    training_dataset = LoggingTensorDataset(torch.from_numpy(train_X).to(device), train_y)
    test_dataset = torch.utils.data.TensorDataset(torch.from_numpy(test_X).to(device), test_y)
    train_loader = torch.utils.data.DataLoader(training_dataset, batch_size=BATCH, shuffle=True)
    test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=len(training_dataset), shuffle=False)
    W0 = rbm.weights.numpy().copy()
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        loss_list, _, nll_list = rbm.train_rbm(train_loader=train_loader, test_loader=test_loader, epochs=EPOCHS,
                                               method="discriminative", generative_factor=0.01)
    n = len(training_dataset)
    order = np.array(training_dataset.log, dtype=np.int16).reshape(EPOCHS, n)
    assert all(sorted(o) == list(range(n)) for o in order)
    pyx = rbm.sample_class_given_x(torch.from_numpy(test_X)).numpy()
    return dict(order=order, loss=np.array(loss_list), nll=np.array(nll_list), acc=np.array(rbm.acc_per_epoch_list),
                auc=np.array(rbm.auc_per_epoch_list), W0_corner=W0[:4, :4].copy(), W=rbm.weights.numpy().copy(),
                b=rbm.visible_bias.numpy().copy(), d=rbm.hidden_bias.numpy().copy(), U=rbm.class_weights.numpy().copy(),
                c=rbm.class_bias.numpy().copy(), pyx_test=pyx)


def main():
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        h9 = dict(seeds=np.array(SEEDS[:3], dtype=np.int64), H=9, lr=LR, epochs=EPOCHS, batch=BATCH)
        for s in SEEDS[:3]:
            r = run(s, 9)
            for k, v in r.items():
                h9[f"s{s}_{k}"] = v
            print("H=9", s, "final acc/auc", r["acc"][-1], r["auc"][-1], flush=True)
        np.savez_compressed(os.path.join(OUT, "rbm_run_breast_h9.npz"), **h9)

        h20 = dict(seeds=np.array(SEEDS, dtype=np.int64), H=20, lr=LR, epochs=EPOCHS, batch=BATCH)
        accs, aucs = [], []
        for s in SEEDS:
            r = run(s, 20)
            for k in ("order", "acc", "auc"):
                h20[f"s{s}_{k}"] = r[k]
            accs.append(r["acc"][-1]); aucs.append(r["auc"][-1])
            print("H=20", s, "final acc/auc", r["acc"][-1], r["auc"][-1], flush=True)
        h20["mean_final_acc"] = np.mean(accs)
        h20["mean_final_auc"] = np.mean(aucs)
        # the row the reference's authors published for this configuration (16 526 parameters):
        # out/BA_data/RBM_results/rbm_performance_summary.csv:2
        h20["published_acc"], h20["published_auc"] = 0.7858974358974359, 0.6610275689223059
        print("H=20 mean over seeds", h20["mean_final_acc"], h20["mean_final_auc"], "published", h20["published_acc"],
              h20["published_auc"])
        np.savez_compressed(os.path.join(OUT, "rbm_run_breast_h20.npz"), **h20)
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
