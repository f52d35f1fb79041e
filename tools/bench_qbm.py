#!/usr/bin/env python
"""Config #5 micro-benchmark: the QUBO Gibbs sampler of Disc_QBM (784v / 256h, B = 73 images, R = 100 reads,
n = 257 unclamped / 256 clamped).  Reports chain-sweeps/s and spin-updates/s of k_qubo_gibbs, the time of one
training minibatch (both phases + fields + statistics), and the oracle (C/OpenMP) on the host cores."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, nargs="+", default=[1, 100, 1000])
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    args = ap.parse_args()
    from qbm_image_classification_amd.sampler import HipQuboOps, auto_beta_range

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(0)
    B, R, H, D = 73, 100, 256, 784
    W_vh = rs.uniform(-1, 1, (1 + D, H)); W_vo = rs.uniform(-1, 1, D); b_h = rs.uniform(-1, 1, H); b_o = rs.uniform(-1, 1, 1)
    W_hh = np.triu(rs.uniform(-1, 1, (H, H)), 1)
    X = (rs.random_sample((B, D)) < 0.13).astype(np.float64)
    Y = rs.randint(0, 2, B)
    out = []
    for clamped in (False, True):
        n = H if clamped else H + 1
        lin = q.linear_fields(X, Y if clamped else None, W_vh, W_vo, b_h, b_o, 1.0)
        J = np.zeros((n, n), np.float32)
        o = 0 if clamped else 1
        if not clamped:
            J[0, 1:] = W_vh[0]; J[1:, 0] = W_vh[0]
        J[o:, o:] += (W_hh + W_hh.T).astype(np.float32)
        Jt = torch.from_numpy(J).cuda()
        b0, b1 = auto_beta_range(lin.cpu().numpy(), J)
        for sw in args.sweeps:
            q.gibbs(lin, Jt, R, sw, b0, b1, 1, 0)
            torch.cuda.synchronize()
            reps = max(1, min(50, int(2000 / max(sw, 1))))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(reps):
                q.gibbs(lin, Jt, R, sw, b0, b1, 1, i)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / reps
            out.append({"phase": "clamped" if clamped else "unclamped", "n": n, "B": B, "R": R, "sweeps": sw, "ms": round(ms, 4),
                        "chain_sweeps_per_s": round(B * R * sw / (ms * 1e-3), 1), "spin_updates_per_s": round(B * R * sw * n / (ms * 1e-3), 1)})
            print(json.dumps(out[-1]), flush=True)
    # CPU oracle on the same unclamped problem, bounded
    from oracle import qbm_oracle as orc

    lin_u = q.linear_fields(X, None, W_vh, W_vo, b_h, b_o, 1.0).cpu().numpy()
    n = H + 1
    J = np.zeros((n, n), np.float32); J[0, 1:] = W_vh[0]; J[1:, 0] = W_vh[0]; J[1:, 1:] += (W_hh + W_hh.T).astype(np.float32)
    b0, b1 = auto_beta_range(lin_u, J)
    sw = 20
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < args.cpu_seconds:
        orc.qubo_gibbs(lin_u, J, R, sw, b0, b1, 1, reps); reps += 1
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"cpu_oracle": True, "cores": len(os.sched_getaffinity(0)), "n": n, "B": B, "R": R, "sweeps": sw, "ms": round(dt * 1e3, 2),
                      "spin_updates_per_s": round(B * R * sw * n / dt, 1)}))


if __name__ == "__main__":
    main()
