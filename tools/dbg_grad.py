import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import synth_inputs, synth_params
from oracle import qbm_oracle as oracle
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet
ops = HipRbmOps("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ops.device)
V, H, C, B = 784, 512, 10, 256
W, b, d, U, c = synth_params(V, H, C, 400 + V + H)
v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 10)
ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, 1, 99, 1000, 512, None, None)
gW, gb, gd, gU, gc = oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
for rep in range(3):
    G = GradSet(V, H, C, ops.device)
    ops.two_phase_grad(V, H, C, t(v_bin), t(y1h), t(ph0), t(vk.astype(np.uint8)), t(yk), t(phk), G)
    for nm, got, want in (("gW", G.gW, gW), ("gb", G.gb, gb), ("gd", G.gd, gd), ("gU", G.gU, gU), ("gc", G.gc, gc)):
        g = got.cpu().numpy()
        bad = np.argwhere(g.view(np.uint32) != np.ascontiguousarray(want, dtype=np.float32).view(np.uint32))
        print(rep, nm, "bad", len(bad), "first", bad[:12].tolist(), "diffs", (g - want)[tuple(bad[:6].T)].tolist() if len(bad) else "")
