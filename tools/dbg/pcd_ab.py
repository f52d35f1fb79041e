"""us per PCD-10 step (784x512, B=256, C=10), eager launches -- for A/B runs of an experiment switch in one gpurun call."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_configs import timeit, params
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps
ops = HipRbmOps("cuda:0")
V, H, C, B, k = 784, 512, 10, 256, 10
P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
v0 = (ops.uniform(1, 0, 0, B, V) < 0.5).to(torch.uint8)
y0 = torch.nn.functional.one_hot(torch.randint(0, C, (B,), device=ops.device), C).float()
ws = ops.cd_workspace(P, B)
cv, cy = v0.clone(), y0.clone()
step = [0]
def f():
    ops.cd_step(P, G, v0, y0, k, 7, step[0] * (3 * k + 1), 0, cv, cy, True, 0.05, 1.0, float(B), 0.0, ws)
    step[0] += 1
print(json.dumps({"env": {k_: v for k_, v in os.environ.items() if k_.startswith("QBM_EXP")}, "pcd10_us": round(timeit(f, reps=300, rounds=4), 2)}))
ops.check_chain_status()
