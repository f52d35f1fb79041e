#!/usr/bin/env python
"""tools/qubo_bench.py -- the Gibbs kernels of the QUBO sampler by themselves (csrc/qubo_kernels.hip) at the shape of
BASELINE.json configs[4]: Disc_QBM 784v / 256h, minibatch of 73 images x 100 reads = 7 300 chains per phase, clamped (n = 256)
and unclamped (n = 257), annealed per image.  One JSON line per sweep count: ms per launch, chain-sweeps/s, spin-updates/s
(SURVEY.md 8.D).  Fields and couplings are those of a Disc_QBM at its initial parameters on synthetic images.

    python tools/qubo_bench.py [--sweeps 100 1000] [--reps 5] [--hidden 256]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def conv_lines(a):
    """HipGibbsSampler.sample_Q (src/qubo/sampler.py:26-33) on the committed conv-QBM QUBOs: ONE QUBO per call, `reads`
    chains, host arrays in and out -- what src/train/pipeline.py would ask of it."""
    import time

    from qbm_image_classification_amd.sampler import HipGibbsSampler

    d = np.load(os.path.join(ROOT, "tests", "golden", "cdqbm_qubo.npz"))
    for key in ("Q_clamped", "Q_unclamped"):
        Q = d[key][0].astype(np.float64)
        n = Q.shape[0]
        for sweeps in a.sweeps:
            smp = HipGibbsSampler(num_sweeps=sweeps, seed=7)
            smp.sparse = not a.dense
            smp.sample_Q(Q, a.reads)
            best = float("inf")
            for _ in range(a.reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s = smp.sample_Q(Q, a.reads)
                best = min(best, time.perf_counter() - t0)
            e = np.einsum("ri,ij,rj->r", s.astype(np.float64), Q, s.astype(np.float64))
            print(json.dumps({"what": f"sample_Q, conv-QBM {key} n = {n} ({int((Q != 0).sum())} non-zeros), {a.reads} reads, {sweeps} sweeps, host call",
                              "kernel": smp.last_kernel, "ms_per_call": round(best * 1e3, 3), "spin_updates_per_s": round(a.reads * sweeps * n / best, 1),
                              "energy_min": round(float(e.min()), 4), "energy_mean": round(float(e.mean()), 4)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, nargs="+", default=[100, 1000])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--images", type=int, default=73)
    ap.add_argument("--reads", type=int, default=100)
    ap.add_argument("--conv", action="store_true", help="sample_Q on the conv-QBM QUBOs of tests/golden/cdqbm_qubo.npz (n = 716 / 717) instead")
    ap.add_argument("--dense", action="store_true", help="--conv: the dense kernel (J rows from L2) instead of the CSR-in-LDS one")
    a = ap.parse_args()
    if a.conv:
        return conv_lines(a)
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    rs = np.random.RandomState(0)
    B, R, H = a.images, a.reads, a.hidden
    X = (rs.random_sample((B, 784)) < 0.13).astype(np.float64)
    Y = rs.randint(0, 2, B)
    np.random.seed(3)
    q = Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=H, seed=5, solver="HIP", sample_count=R, anneal_steps=100)
    Xd, Yd = q._to_dev(X, Y)
    q._push()                              # the parameters as one float64 device buffer
    ops = q.ops
    phases = []
    for clamped in (True, False):
        lin = ops.linear_fields_dev(Xd, Yd if clamped else None, q._theta, 784, H, q.beta_eff)
        J, rowabs, rowmin = q._couplings_dev(clamped)
        phases.append((lin, J, ops.beta_range(lin, rowabs, rowmin)))
    for sweeps in a.sweeps:
        for lin, J, beta in phases:      # warm
            ops.gibbs_dev(lin, J, R, sweeps, beta, 5, 0, 0)
        torch.cuda.synchronize()
        best = [float("inf"), float("inf")]
        for _ in range(a.reps):
            for p, (lin, J, beta) in enumerate(phases):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gibbs_dev(lin, J, R, sweeps, beta, 5, 0, 0)
                e1.record()
                e1.synchronize()
                best[p] = min(best[p], e0.elapsed_time(e1))
        ms = best[0] + best[1]
        spins = B * R * sweeps * (H + H + 1)
        print(json.dumps({"what": f"QUBO Gibbs kernels, Disc_QBM 784v/{H}h, {B} images x {R} reads, {sweeps} sweeps, both phases (n = {H}, {H + 1})",
                          "ms_clamped": round(best[0], 3), "ms_unclamped": round(best[1], 3), "ms_both": round(ms, 3),
                          "chain_sweeps_per_s": round(2 * B * R * sweeps / ms * 1e3, 1), "spin_updates_per_s": round(spins / ms * 1e3, 1)}),
              flush=True)


if __name__ == "__main__":
    main()
