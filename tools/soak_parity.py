#!/usr/bin/env python
"""Soak: many CD-1 steps (784x512, B=256, C=10, uint8 states, fused update) against the oracle, bit for bit, with the
parameters carried from step to step -- an intermittent hazard in the hand-scheduled waits would show up here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import synth_inputs, synth_params
from oracle import qbm_oracle as orc
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ops = HipRbmOps("cuda:0")
K_CD = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # Gibbs iterations per step
PCD = len(sys.argv) > 4 and sys.argv[4] == "pcd"              # persistent chains
if len(sys.argv) > 2:      # 0 / 1 / 2: never / default / always the one-launch chain kernel (include/qbm_hip.h)
    ops.fused_chain_mode(int(sys.argv[2]))
    print("fused chain mode", ops.fused_chain_mode())
V, H, C, B = 784, 512, 10, 256
W, b, d, U, c = synth_params(V, H, C, 1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ops.device)
P = ParamSet(t(W), t(b), t(d), t(U), t(c)); G = GradSet(V, H, C, ops.device)
ws = ops.cd_workspace(P, B)
t0 = time.time()
cv = cy = ocv = ocy = None
if PCD:
    v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 99)
    cv, cy, ocv, ocy = t(v_bin.astype(np.uint8)), t(y1h), v_bin.copy(), y1h.copy()
NOFF = 3 * K_CD + 1
for it in range(steps):
    v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 100 + it)
    out = ops.cd_step(P, G, t(v_bin.astype(np.uint8)), t(y1h), K_CD, 5, NOFF * it, 0, cv, cy, True, 0.05, 1.0, float(B), 0.0001, ws)
    ph0, vk, yk, phk = orc.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, K_CD, 5, NOFF * it, 0, ocv, ocy)
    if PCD:
        vk = ocv
    g = orc.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
    orc.rbm_update(W, b, d, U, c, *g, 0.05, 1.0, float(B), 0.0001)
    ok = (np.array_equal(out[2].cpu().numpy(), vk.astype(np.uint8)) and np.array_equal(ws[4].cpu().numpy().view(np.uint32), phk.view(np.uint32))
          and np.array_equal(P.W.cpu().numpy().view(np.uint32), W.view(np.uint32)) and np.array_equal(P.U.cpu().numpy().view(np.uint32), U.view(np.uint32))
          and np.array_equal(P.b.cpu().numpy().view(np.uint32), b.view(np.uint32)))
    if not ok:
        print(f"MISMATCH at step {it}"); sys.exit(1)
    if it % 50 == 49:
        print(f"{it + 1} steps bit-identical ({time.time() - t0:.0f} s)", flush=True)
ops.check_chain_status(ws)
print("soak ok:", steps, "steps")

# back-to-back determinism: the same 3000 steps twice, launched without host work in between, must end in the same bits
def run_chain(n):
    W0, b0, d0, U0, c0 = synth_params(V, H, C, 2)
    Pn = ParamSet(t(W0), t(b0), t(d0), t(U0), t(c0)); Gn = GradSet(V, H, C, ops.device)
    vb, _, _, _, yh = synth_inputs(B, V, H, C, 7)
    vv, yy = t(vb.astype(np.uint8)), t(yh)
    for it in range(n):
        ops.cd_step(Pn, Gn, vv, yy, 1, 9, 4 * it, 0, None, None, True, 0.05, 1.0, float(B), 0.0, ws)
    torch.cuda.synchronize()
    return Pn.W.clone(), Pn.U.clone(), Pn.d.clone()

a, bb = run_chain(3000), run_chain(3000)
assert all(torch.equal(x, y) for x, y in zip(a, bb)) and torch.isfinite(a[0]).all(), "back-to-back runs differ"
ops.check_chain_status(ws)
print("determinism ok: 2 x 3000 back-to-back steps end in identical parameters")
