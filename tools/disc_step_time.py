"""us per discriminative_training step (qbm_rbm_disc_step_ex through HipRbmOps.disc_step, int32 device labels, eager launches)
at 784 x 512, B = 256 for C = 2, 4, 10 with fp32 and with uint8 inputs: one JSON line per class count."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_configs import timeit, params
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps
ops = HipRbmOps("cuda:0")
for C in (2, 4, 10):
    V, H, B = 784, 512, 256
    P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
    x = ops.uniform(3, 0, 0, B, V) * (ops.uniform(4, 0, 0, B, V) < 0.4)
    lab = torch.randint(0, C, (B,), device=ops.device).to(torch.int32)
    x8 = (ops.uniform(5, 0, 0, B, V) < 0.3).to(torch.uint8)
    print(json.dumps({"C": C, "disc_step_fp32_us": round(timeit(lambda: ops.disc_step(P, G, x, lab, True, True, 0.05, 1.0, float(B), 0.0)), 2),
                      "disc_step_u8_us": round(timeit(lambda: ops.disc_step(P, G, x8, lab, True, True, 0.05, 1.0, float(B), 0.0)), 2)}), flush=True)
