"""k_qubo_gibbs time vs temperature: hot chains flip ~half of the spins per sweep, cold ones almost none --
separates the per-sweep threshold work from the per-flip work (J row read + field update)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from qbm_image_classification_amd.sampler import HipQuboOps, auto_beta_range
q = HipQuboOps("cuda:0")
rs = np.random.RandomState(0)
B, R, n, sw = 73, 100, 257, 200
lin = torch.from_numpy(rs.uniform(-20, 20, (B, n)).astype(np.float32)).cuda()
J = rs.uniform(-1, 1, (n, n)).astype(np.float32); J = np.triu(J, 1); J = J + J.T
Jt = torch.from_numpy(J).cuda()
b0, b1 = auto_beta_range(lin.cpu().numpy(), J)
for name, (x0, x1) in (("auto range", (b0, b1)), ("hot  beta=1e-4", (1e-4, 1e-4)), ("cold beta=1e3", (1e3, 1e3))):
    q.gibbs(lin, Jt, R, sw, x0, x1, 1, 0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(5): st, _ = q.gibbs(lin, Jt, R, sw, x0, x1, 1, i)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:16s} beta {x0:.3g}->{x1:.3g}: {ms:7.2f} ms per {sw} sweeps, {B*R*sw*n/ms/1e6:8.1f} G spin updates/s, mean state {float(st.float().mean()):.3f}")
