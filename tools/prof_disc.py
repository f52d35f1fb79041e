#!/usr/bin/env python
"""Discriminative step only (C from argv) -- run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps
from tools.bench_configs import params

C = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ops = HipRbmOps("cuda:0")
V, H, B = 784, 512, 256
P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
x = ops.uniform(3, 0, 0, B, V) * (ops.uniform(4, 0, 0, B, V) < 0.4)
lab = torch.randint(0, C, (B,), device=ops.device)
for _ in range(200):
    ops.disc_step(P, G, x, lab, True, True, 0.05, 1.0, float(B), 0.0)
torch.cuda.synchronize()
