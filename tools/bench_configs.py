#!/usr/bin/env python
"""Timings of the other BASELINE.json configs and of the remaining hot-path entry points (one JSON line each)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet


def timeit(fn, reps=200, warm=20, rounds=3):
    """best of `rounds` x `reps` eager launches (host hiccups on a shared box otherwise leak into the figure) -> us"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


def params(ops, V, H, C):
    torch.manual_seed(42)
    d = ops.device
    return ParamSet((torch.randn(V, H) * 0.1).to(d), (torch.ones(V) * 0.5).to(d), torch.zeros(H, device=d),
                    (torch.randn(C, H) * 0.1).to(d), torch.zeros(C, device=d))


def main():
    ops = HipRbmOps("cuda:0")
    out = []
    for name, V, H, C, B, k, pcd in (("config#2 784x128 B=128 CD-1", 784, 128, 2, 128, 1, False),
                                     ("config#3 784x512 B=256 PCD-10", 784, 512, 10, 256, 10, True),
                                     ("config#4 shape 784x512 B=256 CD-1 (per GPU)", 784, 512, 10, 256, 1, False),
                                     ("784x512 B=2048 CD-1 (whole config#4 batch on one GPU)", 784, 512, 10, 2048, 1, False)):
        P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
        v0 = (ops.uniform(1, 0, 0, B, V) < 0.5).to(torch.uint8)
        y0 = torch.nn.functional.one_hot(torch.randint(0, C, (B,), device=ops.device), C).float()
        ws = ops.cd_workspace(P, B)
        cv = v0.clone() if pcd else None
        cy = y0.clone() if pcd else None
        step = [0]

        def f():
            ops.cd_step(P, G, v0, y0, k, 7, step[0] * (3 * k + 1), 0, cv, cy, True, 0.05, 1.0, float(B), 0.0, ws)
            step[0] += 1
        us = timeit(f)
        out.append({"what": name, "us_per_step": round(us, 2), "gibbs_samples_per_s": round(B * k / us * 1e6, 1),
                    "visible_unit_updates_per_s": round(B * k * V / us * 1e6, 1)})
        print(json.dumps(out[-1]), flush=True)
    for C in (2, 10):
        V, H, B = 784, 512, 256
        P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
        x = ops.uniform(3, 0, 0, B, V) * (ops.uniform(4, 0, 0, B, V) < 0.4)
        lab = torch.randint(0, C, (B,), device=ops.device)
        us = timeit(lambda: ops.disc_step(P, G, x, lab, True, True, 0.05, 1.0, float(B), 0.0))
        out.append({"what": f"discriminative_training step 784x512 B=256 C={C} (reference CPU: 167/235 ms)", "us_per_step": round(us, 2)})
        print(json.dumps(out[-1]), flush=True)
        if C == 10:
            x8 = (ops.uniform(5, 0, 0, B, V) < 0.3).to(torch.uint8)
            us8 = timeit(lambda: ops.disc_step(P, G, x8, lab, True, True, 0.05, 1.0, float(B), 0.0))
            out.append({"what": f"discriminative_training step 784x512 B=256 C={C}, binary inputs as uint8", "us_per_step": round(us8, 2)})
            print(json.dumps(out[-1]), flush=True)
        us = timeit(lambda: ops.free_energy(P, x))
        out.append({"what": f"sample_class_given_x 784x512 B=256 C={C} (reference CPU: 29/108 ms)", "us_per_call": round(us, 2)})
        print(json.dumps(out[-1]), flush=True)
    # a whole epoch from the device-resident dataset: permutation + one gather launch per step, no host copies
    from qbm_image_classification_amd.device_data import DeviceDataset

    V, H, C, B, N = 784, 512, 10, 256, 60160
    rs = np.random.RandomState(0)
    ds = DeviceDataset((rs.random_sample((N, V)) < 0.2).astype(np.uint8), rs.randint(0, C, N), B, C, storage="bits", seed=1)
    P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device); ws = ops.cd_workspace(P, B)
    dt = float("inf")
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k, (xb, lab, one) in enumerate(ds.batches(rep, True)):
            ops.cd_step(P, G, xb, one, 1, 7, (rep * len(ds) + k) * 4, 0, None, None, True, 0.05, 1.0, float(B), 0.0, ws)
        torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
    out.append({"what": f"epoch of {N} bit-packed images ({ds.bytes_resident() / 1e6:.1f} MB resident), CD-1 784x512 B=256, "
                        "device permutation + one gather launch per epoch", "ms_per_epoch": round(dt * 1e3, 2),
                "us_per_step": round(dt / len(ds) * 1e6, 2), "images_per_s": round(N / dt, 1)})
    print(json.dumps(out[-1]), flush=True)
    # greedy k-center coreset at the size config #5 draws from (MNIST 0/1 train split, 256-d hidden means)
    from qbm_image_classification_amd import coreset

    emb = torch.rand(12665, 256, device=ops.device)
    for fused, how in ((True, "all picks in ONE launch: 256 persistent workgroups, rows in LDS, one granule per pick"), (False, "one launch per pick")):
        coreset.kcenter_picks(emb, [0], 10, fused=fused)
        dt = float("inf")
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            coreset.kcenter_picks(emb, [0], 1001, fused=fused)
            torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
        out.append({"what": f"k_center_greedy 12665 x 256 embeddings, 1000 centres ({how})", "ms": round(dt * 1e3, 2),
                    "us_per_pick": round(dt / 1001 * 1e6, 2)})
        print(json.dumps(out[-1]), flush=True)
    # one Disc_QBM training minibatch at config #5 (both phases, fields, statistics, host update)
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    rs = np.random.RandomState(0)
    X = (rs.random_sample((73, 784)) < 0.13).astype(np.float64); Y = rs.randint(0, 2, 73)
    for sweeps in (100, 1000):
        np.random.seed(1)
        q = Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=256, seed=3, solver="HIP", sample_count=100, anneal_steps=sweeps)
        q.train_for_one_iteration(X, Y, 0.01)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            q.train_for_one_iteration(X, Y, 0.01)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        out.append({"what": f"Disc_QBM.train_for_one_iteration 784v/256h B=73 R=100 sweeps={sweeps} (146 QUBOs)", "ms": round(dt * 1e3, 2),
                    "ms_per_qubo": round(dt * 1e3 / 146, 3), "chain_sweeps_per_s": round(2 * 73 * 100 * sweeps / dt, 1),
                    "spin_updates_per_s": round(73 * 100 * sweeps * (256 + 257) / dt, 1)})
        print(json.dumps(out[-1]), flush=True)
    # sample_Q (src/qubo/sampler.py:26-33) on a committed conv-QBM QUBO: the sparse-coupling kernel, one QUBO per host call
    import os

    from qbm_image_classification_amd.sampler import HipGibbsSampler

    Q = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cdqbm_qubo.npz"))["Q_unclamped"][0]
    smp = HipGibbsSampler(num_sweeps=100, seed=7)
    smp.sample_Q(Q, 100)
    dt = float("inf")
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        smp.sample_Q(Q, 100)
        dt = min(dt, time.perf_counter() - t0)
    out.append({"what": f"sample_Q conv-QBM QUBO n = {Q.shape[0]}, 100 reads x 100 sweeps, host call ({smp.last_kernel} kernel)", "ms": round(dt * 1e3, 3)})
    print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
