#!/usr/bin/env python
"""rbm_main.py -- drop-in for the reference's rbm_main.py on the HIP path.

Same command line (rbm_main.py:12-30 of the reference), same seed list (:32-33), same flow (:36-82):
load BreastMNIST -> flatten -> ClassificationRBM -> train_rbm -> pickle acc/auc per epoch.  Differences:
`--type generative|hybrid` work (CD-k / PCD on the GPU; the reference raises NotImplementedError), the
output directory is created (the reference crashes if it is missing), and the data
are looked up under --data-root (default: the reference layout src/data/..., else the committed fixture
tests/golden/data_breastmnist.npz).  Extra flags: --seeds N, --binarise, --persistent, --bars-and-stripes."""
import argparse
import os
import pickle

import numpy as np
import torch

from qbm_image_classification_amd import data_loader
from qbm_image_classification_amd.classification_rbm import ClassificationRBM
from qbm_image_classification_amd.launch import finalize_distributed, init_distributed

RANK, WORLD, DEVICE = init_distributed()   # (0, 1, "cuda:0") unless launched by torch.distributed.run

USE_GPU = True

# The reference's command line (flag, type, default: rbm_main.py:12-30 there), then the build-side extras.
REFERENCE_FLAGS = [
    ("--lr", float, 0.08714599435919934, "learning rate"),
    ("--epochs", int, 20, "training epochs"),
    ("--batch-size", int, 7, "minibatch size"),
    ("--early-stop", int, 15, "accepted for compatibility (the reference never reads it either)"),
    ("--visible-units", int, 784, "input dimension"),
    ("--hidden-units", int, 9, "hidden units"),
    ("--cd-k", int, 1, "Gibbs steps of the contrastive-divergence chain"),
    ("--type", str, "discriminative", "training rule: discriminative, hybrid or generative"),
    ("--sparsity-coeffectient", float, 0.00, "subtracted from the biases after every update"),
    ("--data-folder", str, "data", "accepted for compatibility"),
    ("--generative-factor", float, 0.01, "weight of the CD-k step in hybrid training"),
]
EXTRA_FLAGS = [
    ("--data-root", str, None, "directory holding src/data/medmnist/breastmnist.npz"),
    ("--seeds", int, 10, "how many of the 10 reference seeds to run"),
    ("--out-dir", str, "Breast_beta_hyper_RBM", "where acc/auc per epoch are pickled (created if missing)"),
]
SWITCHES = [
    ("--no-cuda", "accepted for compatibility: the HIP path has no CPU fallback"),
    ("--binarise", "threshold the images at 0.5 ({0,1} visibles, stored bit-packed on the device)"),
    ("--persistent", "PCD instead of CD-k for generative / hybrid training"),
    ("--bars-and-stripes", "4x4 Bars-and-Stripes instead of BreastMNIST (16 visible units)"),
    ("--host-loader", "torch DataLoader(shuffle=True) as in the reference instead of the device-resident dataset"),
]
parser = argparse.ArgumentParser(description="ClassificationRBM on the MI355X HIP path (the reference's rbm_main.py command line)")
for flag, typ, default, text in REFERENCE_FLAGS + EXTRA_FLAGS:
    parser.add_argument(flag, type=typ, default=default, help=text)
for flag, text in SWITCHES:
    parser.add_argument(flag, action="store_true", help=text)


def find_breastmnist(root):
    here = os.path.dirname(os.path.abspath(__file__))
    for cand in ([os.path.join(root, 'src/data/medmnist/breastmnist.npz')] if root else []) + [
            'src/data/medmnist/breastmnist.npz', os.path.join(here, 'tests/golden/data_breastmnist.npz')]:
        if os.path.exists(cand):
            return cand
    raise FileNotFoundError('breastmnist.npz not found (use --data-root)')


def main(args):
    seeds = [1967690937, 2286980494, 3620295971, 1662044193, 1825595160, 3054779705, 900327972, 1620954898,
             3699850877, 3492574433][:args.seeds]
    os.makedirs(args.out_dir, exist_ok=True)
    results = []
    for seed in seeds:
        np.random.seed(seed)
        torch.manual_seed(seed)
        print("Loading data...")
        if args.bars_and_stripes:
            x, y = data_loader.bars_and_stripes(4)
            train_X = test_X = x.astype(np.float32)
            train_y = test_y = y
            args.visible_units = 16
        else:
            (train_X, train_y), (val_X, val_y), (test_X, test_y) = data_loader.get_medmnist(find_breastmnist(args.data_root))
            print("Data loaded")
            print("Train train_x shape: ", train_X.shape)
            train_X, val_X, test_X = data_loader.preprocess_images(train_X, val_X, test_X)
        if args.binarise:
            train_X = data_loader.binarise(train_X).astype(np.float32)
            test_X = data_loader.binarise(test_X).astype(np.float32)

        rbm = ClassificationRBM(args.visible_units, args.hidden_units, args.cd_k, num_classes=2, learning_rate=args.lr,
                                sparse_constant=args.sparsity_coeffectient, use_cuda=not args.no_cuda, seed=seed,
                                persistent=args.persistent, device=DEVICE)
        device = rbm.get_device(False)   # host tensors; the class stages minibatches to the GPU
        train_y_t = torch.from_numpy(np.array(train_y)).to(device).long()
        test_y_t = torch.from_numpy(np.array(test_y)).to(device).long()
        training_dataset = torch.utils.data.TensorDataset(torch.from_numpy(train_X).to(device), train_y_t)
        test_dataset = torch.utils.data.TensorDataset(torch.from_numpy(test_X).to(device), test_y_t)
        if args.host_loader:   # the reference's loader: host-side shuffling and slicing every step
            train_loader = torch.utils.data.DataLoader(training_dataset, batch_size=args.batch_size, shuffle=True)
        else:                  # default: the set lives on the GPU, permutation + gather happen there
            from qbm_image_classification_amd.device_data import DeviceDataset

            train_loader = DeviceDataset(train_X, np.array(train_y), args.batch_size, num_classes=2,
                                         storage="bits" if args.binarise else "f32", seed=seed, device=str(rbm.device),
                                         rank=RANK, world=WORLD)   # under torchrun: --batch-size rows per rank
        test_loader = torch.utils.data.DataLoader(test_dataset, batch_size=len(training_dataset), shuffle=False)

        loss_list, best_model, nll_list = rbm.train_rbm(train_loader=train_loader, test_loader=test_loader,
                                                        epochs=args.epochs, method=args.type,
                                                        generative_factor=args.generative_factor)
        if RANK == 0:   # the replicas are identical: one of them writes
            with open(f'{args.out_dir}/acc_per_epoch{seed}.pkl', 'wb') as f:
                pickle.dump(rbm.acc_per_epoch_list, f)
            with open(f'{args.out_dir}/auc_per_epoch{seed}.pkl', 'wb') as f:
                pickle.dump(rbm.auc_per_epoch_list, f)
        results.append((seed, rbm.acc_per_epoch_list[-1], rbm.auc_per_epoch_list[-1]))
    print("seed, final test accuracy, final test AUC")
    for r in results:
        print(*r)
    return results


if __name__ == '__main__':
    try:
        main(parser.parse_args())
    finally:
        finalize_distributed()
