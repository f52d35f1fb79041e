#!/usr/bin/env python
"""qbm_main.py -- drop-in for the reference's qbm_main.py with the QUBO sampler on the GPU (solver "HIP").

Same `main(...)` signature and command line as the reference (qbm_main.py:23-25, :138-216).  Differences: the
default solver is "HIP" ("SA" is accepted as an alias), `type=bool` flags are parsed as real booleans (the
reference turns any non-empty string into True), the results are printed and
pickled by this script instead of the reference's metrics.save_result (which raises AttributeError for
Disc_QBM, Q12), and datasets are looked up under --data_root (default: the reference layout src/data/...,
else the committed fixtures under tests/golden/)."""
import argparse
import os
import pickle
import random

import numpy as np

from qbm_image_classification_amd import data_loader, metrics
from qbm_image_classification_amd.disc_qbm import Disc_QBM
from qbm_image_classification_amd.launch import finalize_distributed, init_distributed

RANK, WORLD, DEVICE = init_distributed()   # (0, 1, "cuda:0") unless launched by torch.distributed.run

HERE = os.path.dirname(os.path.abspath(__file__))


def _find(root, rel, fixture):
    for cand in ([os.path.join(root, rel)] if root else []) + [rel, os.path.join(HERE, 'tests/golden', fixture)]:
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError(f'{rel} not found (use --data_root)')


def load(data_set, data_root):
    if data_set == "breastmnist":
        return data_loader.get_medmnist(_find(data_root, 'src/data/medmnist/breastmnist.npz', 'data_breastmnist.npz'))
    if data_set == "pneumoniamnist":
        return data_loader.get_medmnist(_find(data_root, 'src/data/medmnist/pneumoniamnist.npz', 'data_pneumoniamnist.npz'))
    if data_set == "mnist":
        # MNIST train images are absent from the reference checkout; the fixture holds the binarised 0/1 test digits
        f = np.load(os.path.join(HERE, 'tests/golden/data_mnist01_t10k_bits.npz'))
        x = data_loader.unpack_bits(f['bits'], 784).astype(np.float32).reshape(-1, 28, 28)
        y = f['labels'].astype(np.uint8)
        n = int(0.8 * len(x))
        return (x[:n], y[:n]), (x[n:], y[n:]), (x[n:], y[n:])
    raise ValueError("Invalid dataset")


def main(seed=19, n_hidden_nodes=10, solver="HIP", sample_count=100, anneal=1000, beta_eff=1.0, epochs=3, batch_size=10,
         learning_rate=0.01, restricted=True, data_set="mnist", num_classes=2, parallelize=False, save="", name="",
         data_root=None, max_train=None):
    print("Start")
    random.seed(seed)
    np.random.seed(seed)
    print("Seed is " + str(seed))
    print("Loading data...")
    (train_X, train_y), (val_X, val_y), (test_X, test_y) = load(data_set, data_root)
    print("Data loaded")
    train_X, test_X, val_X = data_loader.preprocess_images(train_X, test_X, val_X)
    if max_train:
        train_X, train_y = train_X[:max_train], train_y[:max_train]
    param_string = "_se" + str(seed) + "_h" + str(n_hidden_nodes) + "_sol" + solver + "_sc" + str(sample_count) + "_b" + str(
        beta_eff) + "_e" + str(epochs) + "_bs" + str(batch_size) + "_l" + str(learning_rate) + "_r" + str(restricted) + \
        "_data" + data_set + "_n_" + name
    print(param_string)
    print('Creating QBM...')
    dqbm = Disc_QBM(seed=seed, epochs=epochs, n_hidden_nodes=n_hidden_nodes, num_classes=num_classes, solver=solver,
                    sample_count=sample_count, anneal_steps=anneal, beta_eff=beta_eff, restricted=restricted,
                    param_string=param_string, speicherort=save, dim_input=784, parallelize=parallelize,
                    use_one_hot_encoding=False, device=DEVICE)   # under torchrun every minibatch is sharded over the ranks
    print('QBM created')
    print('Training QBM...')
    dqbm.train_model(train_X, train_y, test_X, test_y, batch_size=batch_size, learning_rate=learning_rate)
    print('QBM trained')
    print("Predict on test data...")
    predictions = dqbm.predict_batch(test_X)
    acc, f1, precision, recall, auc = metrics.get_metrics(np.asarray(test_y), predictions, ["healthy", "pneumonia"])
    if RANK == 0:
        with open((save or "") + name + "acc_auc.pkl", "wb") as f:
            pickle.dump((acc, auc), f)
        metrics.save_history((save or "") + name, dqbm.training_history)
    print("Accuracy: ", acc)
    print("F1 Score: ", f1)
    print("Precision: ", precision)
    print("Recall: ", recall)
    print("AUC Score: ", auc)
    return acc, auc


def _bool(s):
    return str(s).lower() in ("1", "true", "yes", "y")


if __name__ == '__main__':
    # the reference's flags (qbm_main.py:138-216 there: short name, long name, type, default), then the build-side extras
    CLI = [
        ("-hn", "--hnodes", int, 2, "hidden units"),
        ("-lr", "--learning_rate", float, 0.4529451796571889, "learning rate"),
        ("-r", "--restricted", _bool, False, "no couplings between hidden units"),
        ("-e", "--epochs", int, 20, "training epochs"),
        ("-b", "--batch_size", int, 73, "minibatch size"),
        ("-s", "--seed", int, 3492574433, "seed of the numpy / sampler streams"),
        ("-sc", "--sample_count", int, 100, "reads per QUBO"),
        (None, "--solver", str, "HIP", "'HIP' (GPU Gibbs / annealing); 'SA' is accepted as an alias"),
        (None, "--data_set", str, "pneumoniamnist", "'mnist', 'breastmnist' or 'pneumoniamnist'"),
        (None, "--num_classes", int, 2, "classes in the dataset"),
        (None, "--parallelize", _bool, True, "accepted for compatibility: the GPU path is always batched"),
        (None, "--load_path", str, "out/", "output directory (created if missing)"),
        (None, "--name", str, "qucun_3492574433", "name of the run"),
        (None, "--anneal", int, 1000, "sweeps per read"),
        (None, "--data_root", str, None, "directory holding the reference's src/data tree"),
        (None, "--max_train", int, None, "use only the first N training images"),
    ]
    parser = argparse.ArgumentParser(description="Discriminative QBM with the QUBO sampler on the GPU")
    for short, long_, typ, default, text in CLI:
        parser.add_argument(*([short] if short else []), long_, type=typ, default=default, help=text)
    flags = parser.parse_args()
    print("Running with solver", flags.solver)
    os.makedirs(flags.load_path, exist_ok=True)
    try:
        main(epochs=flags.epochs, n_hidden_nodes=flags.hnodes, learning_rate=flags.learning_rate, batch_size=flags.batch_size,
             solver=flags.solver, restricted=flags.restricted, seed=flags.seed % (2 ** 32), data_set=flags.data_set,
             num_classes=flags.num_classes, parallelize=flags.parallelize, sample_count=flags.sample_count,
             save=flags.load_path, name=flags.name, anneal=flags.anneal, data_root=flags.data_root, max_train=flags.max_train)
    finally:
        finalize_distributed()
