import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import synth_inputs, synth_params
from oracle import qbm_oracle as orc
from qbm_image_classification_amd.rbm_ops import HipRbmOps, ParamSet
orc.build()
ops = HipRbmOps("cuda:0")
V, H, C, B = 16, 8, 2, 6
W, b, d, U, c = synth_params(V, H, C, 100 + V + H)
v_bin, x_grey, h_bin, labels, y1h = synth_inputs(B, V, H, C, 7)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
P = ParamSet(t(W), t(b), t(d), t(U), t(c))
p, _ = ops.sample_hidden(P, t(v_bin), t(y1h))
po, _ = orc.rbm_sample_hidden(v_bin, y1h, W, d, U, sample=False)
pn, _ = orc.rbm_sample_hidden(v_bin, None, W, d, U, sample=False)
ysw = y1h[[1, 0, 3, 2, 5, 4]]
ps, _ = orc.rbm_sample_hidden(v_bin, ysw, W, d, U, sample=False)
p = p.cpu().numpy()
print("labels", labels)
print("max|gpu-oracle| per row", np.abs(p - po).max(1))
print("max|gpu-(no class)| per row", np.abs(p - pn).max(1))
print("max|gpu-(y of neighbour row)| per row", np.abs(p - ps).max(1))
