#!/usr/bin/env python
"""Soak of the QUBO Gibbs kernels against the oracle, bit for bit: random sizes (1 .. 1024 variables: every tile count, the
n = 64 k + 1 form, the LDS triangle and the L2 reader), chain counts (both sides of the kernels' launch choices), sweeps,
annealing ranges, dense and sparse couplings (the CSR kernel with and without its helper wavefront).
    python tools/soak_sampler.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import qbm_oracle as orc
from qbm_image_classification_amd.sampler import HipQuboOps

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
q = HipQuboOps("cuda:0")
rs = np.random.RandomState(int(os.environ.get("SOAK_SEED", "1")))
t0, cases, csr_cases = time.time(), 0, 0
while time.time() - t0 < budget:
    kind = rs.randint(0, 4)
    n = int(rs.choice([rs.randint(1, 65), rs.randint(65, 277), rs.randint(277, 1025), 64 * rs.randint(1, 6) + 1]))
    B = int(rs.choice([1, 2, 3, rs.randint(1, 40)]))
    R = int(rs.choice([1, 5, rs.randint(1, 120)]))
    while B * R * n > 400000:          # (the oracle walks every row of J per flip)
        R = max(1, R // 2); B = max(1, B - 1)
    sweeps = int(rs.choice([0, 1, 2, rs.randint(1, 8)]))
    b0 = float(rs.choice([1.0, 0.05, rs.uniform(0.01, 2.0)])); b1 = float(rs.choice([b0, b0 * rs.uniform(1.0, 50.0)]))
    seed, off, chain0 = int(rs.randint(0, 1 << 30)), int(rs.randint(0, 1 << 20)), int(rs.choice([0, rs.randint(0, 5000)]))
    lin = (rs.standard_normal((B, n)) * rs.choice([0.1, 1.0, 3.0])).astype(np.float32)
    if kind == 0 or n < 3:             # dense couplings
        Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * np.float32(rs.choice([0.05, 0.3])), 1)
    else:                              # sparse
        Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * (rs.random_sample((n, n)) < rs.choice([0.01, 0.05])), 1)
    J = (Ju + Ju.T).astype(np.float32)
    lt, Jt = torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda()
    so, mo = orc.qubo_gibbs(lin, J, R, sweeps, b0, b1, seed, off, chain0)
    st, mean = q.gibbs(lt, Jt, R, sweeps, b0, b1, seed, off, True, chain0)
    assert np.array_equal(st.cpu().numpy(), so) and np.array_equal(mean.cpu().numpy(), mo), ("dense kernel", n, B, R, sweeps, b0, b1, seed, off, chain0)
    if kind != 0:
        sc, mc, head = q.gibbs_csr(lt, Jt, R, sweeps, b0, b1, seed, off, True, chain0)
        if int(head[0]) == 0:
            assert np.array_equal(sc.cpu().numpy(), so) and np.array_equal(mc.cpu().numpy(), mo), ("csr kernel", n, B, R, sweeps, b0, b1, seed, off, chain0)
            csr_cases += 1
    cases += 1
    if cases % 50 == 0:
        print(f"{cases} cases ({csr_cases} through the CSR kernel), {time.time() - t0:.0f} s", flush=True)
print(f"soak ok: {cases} cases, {csr_cases} through the CSR kernel, all bit-identical to the oracle")
