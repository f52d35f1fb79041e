#!/usr/bin/env python
"""tools/bench_train_loop.py -- throughput of the DROP-IN loops next to the raw C-ABI step loops.

The reference's hot loops are ClassificationRBM.train_rbm (src/ClassificationRBM.py:167-200) and Disc_QBM.train_model
(src/model/faster_dqbm.py:1094-1150); bench.py and tools/bench_configs.py time the C-ABI step underneath them.  This tool
times both sides on the same data so that what the Python loop adds is visible: one JSON line per configuration.

  RBM   784 x 512, C = 10, B = 256, 60 160 bit-packed images resident in HBM (DeviceDataset), one epoch:
        train_rbm(method = discriminative | generative (CD-1) | generative (PCD-10)) vs a loop of ops.disc_step / ops.cd_step
        over the same minibatches.
  QBM   784v / 256h, B = 73, R = 100 reads, 100 sweeps, 20 minibatches: Disc_QBM.train_model (one epoch, validation on 73
        images) vs a loop of Disc_QBM._step_device.

    python tools/bench_train_loop.py [--images N] [--quick]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from qbm_image_classification_amd.classification_rbm import ClassificationRBM  # noqa: E402
from qbm_image_classification_amd.device_data import DeviceDataset  # noqa: E402
from qbm_image_classification_amd.rbm_ops import GradSet  # noqa: E402

V, H, C, B = 784, 512, 10, 256


def best_of(fn, reps):
    best = float("inf")
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def rbm_lines(n_images, reps):
    rs = np.random.RandomState(0)
    x = (rs.random_sample((n_images, V)) < 0.2).astype(np.uint8)
    y = rs.randint(0, C, n_images)
    for method, k, pcd in (("discriminative", 1, False), ("generative", 1, False), ("generative", 10, True)):
        ds = DeviceDataset(x, y, B, C, storage="bits", seed=1)
        rbm = ClassificationRBM(V, H, k, num_classes=C, learning_rate=0.05, seed=42, persistent=pcd, device="cuda:0")
        n_steps = len(ds)

        def drop_in():
            with contextlib.redirect_stdout(io.StringIO()):
                rbm.train_rbm(ds, epochs=1, method=method)

        drop_in()                                                     # warm-up epoch (workspaces, code objects)
        t_loop = best_of(drop_in, reps)

        # the raw loop: the same minibatches, the C-ABI step only (what bench.py / tools/bench_configs.py time)
        ops, P, G = rbm.ops, rbm._params(), GradSet(V, H, C, rbm.device)
        cv = cy = None
        if pcd:
            cv = torch.zeros(B, V, dtype=torch.uint8, device=rbm.device)
            cy = torch.zeros(B, C, device=rbm.device)
        ep = [100]

        def raw():
            ep[0] += 1
            for i, (xb, lab, one) in enumerate(ds.batches(ep[0], method != "discriminative")):
                n = xb.shape[0]
                if method == "discriminative":
                    ops.disc_step(P, G, xb, lab, True, True, 0.05, 1.0, float(n), 0.0)
                else:
                    ops.cd_step(P, G, xb, one, k, 42, (3 * k + 1) * i, 0, None if cv is None else cv[:n], None if cy is None else cy[:n],
                                True, 0.05, 1.0, float(n), 0.0)

        raw()
        t_raw = best_of(raw, reps)
        ops.check_chain_status()
        # ... and the same loop with the training loop's monitor riding in the step (what train_rbm asks of the C ABI): the
        # difference to `raw` is the monitor's device cost, the difference to train_rbm is Python
        t_mon = None
        if method != "discriminative":
            sums = torch.zeros(2, dtype=torch.float64, device=rbm.device)

            def raw_mon():
                ep[0] += 1
                for i, (xb, lab, one) in enumerate(ds.batches(ep[0], True)):
                    n = xb.shape[0]
                    ops.cd_step(P, G, xb, one, k, 42, (3 * k + 1) * i, 0, None if cv is None else cv[:n], None if cy is None else cy[:n],
                                True, 0.05, 1.0, float(n), 0.0, monitor=(lab, sums))

            raw_mon()
            t_mon = best_of(raw_mon, reps)
            ops.check_chain_status()
        # from a device-resident dataset the discriminative rule and CD-1 replay their epochs from ONE captured hipGraph (the
        # second epoch of a fresh object captures it) -- 3 epochs per call, per-epoch time
        t_graph = None
        if method == "discriminative" or (k == 1 and not pcd):
            def three():
                with contextlib.redirect_stdout(io.StringIO()):
                    rbm.train_rbm(ds, epochs=3, method=method)
            three()
            t_graph = best_of(three, reps) / 3
        name = method + (f" PCD-{k}" if pcd else (f" CD-{k}" if method == "generative" else ""))
        print(json.dumps({"what": f"train_rbm {name}, 784x512 C=10 B=256, {n_images} bit-packed images from DeviceDataset, one epoch",
                          "steps": n_steps, "train_rbm_ms_per_epoch": round(t_loop * 1e3, 2), "train_rbm_us_per_step": round(t_loop / n_steps * 1e6, 2),
                          "train_rbm_images_per_s": round(n_images / t_loop, 1), "raw_step_loop_ms_per_epoch": round(t_raw * 1e3, 2),
                          "raw_us_per_step": round(t_raw / n_steps * 1e6, 2), "raw_images_per_s": round(n_images / t_raw, 1),
                          "raw_with_monitor_us_per_step": None if t_mon is None else round(t_mon / n_steps * 1e6, 2),
                          "train_rbm_graph_replayed_epoch_us_per_step": None if t_graph is None else round(t_graph / n_steps * 1e6, 2),
                          "train_rbm_over_raw": round(t_loop / t_raw, 3)}), flush=True)


def dp_line(n_images, reps):
    """The data-parallel code path of the PRODUCT loop on ONE rank (no multi-GPU box is available to the build: N > 1 is
    unmeasured on hardware): ClassificationRBM(data_parallel=True).train_rbm under a 1-rank RCCL group -- per step the chain,
    the gradient launch writing its row count, ONE all-reduce of the flat gradient buffer, and qbm_rbm_update_dev reading the
    global row count on the device (no host round trip)."""
    import torch.distributed as dist

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        rs = np.random.RandomState(0)
        x = (rs.random_sample((n_images, V)) < 0.2).astype(np.uint8)
        y = rs.randint(0, C, n_images)
        for method in ("generative", "discriminative"):
            ds = DeviceDataset(x, y, B, C, storage="bits", seed=1)
            rbm = ClassificationRBM(V, H, 1, num_classes=C, learning_rate=0.05, seed=42, device="cuda:0", data_parallel=True)

            def drop_in():
                with contextlib.redirect_stdout(io.StringIO()):
                    rbm.train_rbm(ds, epochs=1, method=method)

            drop_in()
            t = best_of(drop_in, reps)          # a call of ONE epoch never captures: eager launches + one collective per step

            # from its second epoch on the loop replays the epoch from ONE hipGraph with the RCCL all-reduce and the device-count
            # update inside (round 4): 3 epochs per call, per-epoch time
            def three():
                with contextlib.redirect_stdout(io.StringIO()):
                    rbm.train_rbm(ds, epochs=3, method=method)
            three()
            t_graph = best_of(three, reps) / 3
            print(json.dumps({"what": f"train_rbm {method}{' CD-1' if method == 'generative' else ''}, DATA-PARALLEL code path on ONE rank (1-rank RCCL "
                                      f"group; N > 1 unmeasured on hardware), 784x512 C=10 B=256, {n_images} images, one epoch",
                              "steps": len(ds), "eager_ms_per_epoch": round(t * 1e3, 2), "eager_us_per_step": round(t / len(ds) * 1e6, 2),
                              "graph_replayed_epoch_us_per_step": round(t_graph / len(ds) * 1e6, 2), "epoch_graph_captured": rbm._epoch_graph is not None,
                              "train_rbm_images_per_s": round(n_images / t_graph, 1)}), flush=True)
    finally:
        dist.destroy_process_group()


def qbm_lines(reps, n_batches=20, sweeps=100):
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    rs = np.random.RandomState(0)
    Bq, R = 73, 100
    X = (rs.random_sample((Bq * n_batches, 784)) < 0.13).astype(np.float64)
    Y = rs.randint(0, 2, Bq * n_batches)
    np.random.seed(1)
    q = Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=256, seed=3, solver="HIP", sample_count=R, anneal_steps=sweeps, epochs=1,
                 speicherort="/tmp/qbm_bench_", param_string="loop")

    def drop_in():
        with contextlib.redirect_stdout(io.StringIO()):
            q.train_model(X, Y, X[:Bq], Y[:Bq], batch_size=Bq, learning_rate=0.01)

    drop_in()
    t_loop = best_of(drop_in, reps)
    Xd, Yd = q._to_dev(X, Y)

    def raw():
        for b in range(0, Xd.shape[0], Bq):
            q._step_device(Xd[b:b + Bq], Yd[b:b + Bq], 0.01)
        q.predict_batch(Xd[:Bq])                                      # train_model's per-epoch validation

    raw()
    t_raw = best_of(raw, reps)
    spin = 2 * Bq * R * sweeps * 256.5 * n_batches                     # two phases, n = 256 / 257
    print(json.dumps({"what": f"Disc_QBM.train_model 784v/256h, B={Bq}, R={R}, {sweeps} sweeps, {n_batches} minibatches + validation of {Bq} images, one epoch",
                      "train_model_ms_per_epoch": round(t_loop * 1e3, 2), "train_model_ms_per_minibatch": round(t_loop / n_batches * 1e3, 3),
                      "raw_step_loop_ms_per_epoch": round(t_raw * 1e3, 2), "raw_ms_per_minibatch": round(t_raw / n_batches * 1e3, 3),
                      "train_model_over_raw": round(t_loop / t_raw, 3), "spin_updates_per_s": round(spin / t_raw, 1),
                      "chain_sweeps_per_s": round(2 * Bq * R * sweeps * n_batches / t_raw, 1)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=60160)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--quick", action="store_true", help="6 016 images, one repetition")
    ap.add_argument("--no-qbm", action="store_true")
    ap.add_argument("--no-dp", action="store_true", help="skip the data-parallel code path on one rank")
    a = ap.parse_args()
    if a.quick:
        a.images, a.reps = 6016, 1
    rbm_lines(a.images, a.reps)
    if not a.no_qbm:
        qbm_lines(a.reps)
    if not a.no_dp:
        dp_line(a.images, a.reps)
