"""What a sweep of the sparse sampler costs WITHOUT flips (frozen chains: beta = 1e4): thresholds, tile tests, loop."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from qbm_image_classification_amd.sampler import HipGibbsSampler
d = np.load(os.path.join(ROOT, "tests", "golden", "cdqbm_qubo.npz"))
Q = d["Q_unclamped"][0].astype(np.float64)
for br in (None, (1e4, 1e4)):
    for sparse in (True, False):
        smp = HipGibbsSampler(num_sweeps=1000, seed=7, beta_range=br)
        smp.sparse = sparse
        smp.sample_Q(Q, 100)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); smp.sample_Q(Q, 100); best = min(best, time.perf_counter() - t0)
        print(json.dumps({"beta_range": br, "kernel": smp.last_kernel, "ms_per_1000_sweeps": round(best * 1e3, 3)}), flush=True)
