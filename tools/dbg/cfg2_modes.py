"""config #2 (784x128, B=128, CD-1) and a few small shapes under the launch-per-phase path (mode 1 at k = 1) and the one-launch
chain (mode 2): us per step, eager launches (tools/bench_configs.timeit)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_configs import timeit, params
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps
ops = HipRbmOps("cuda:0")
for V, H, C, B in ((784, 128, 2, 128), (784, 128, 2, 256), (784, 256, 2, 128), (784, 512, 10, 128)):
    res = {}
    for mode in (1, 2):
        ops.fused_chain_mode(mode)
        P = params(ops, V, H, C); G = GradSet(V, H, C, ops.device)
        v0 = (ops.uniform(1, 0, 0, B, V) < 0.5).to(torch.uint8)
        y0 = torch.nn.functional.one_hot(torch.randint(0, C, (B,), device=ops.device), C).float()
        ws = ops.cd_workspace(P, B)
        step = [0]
        def f():
            ops.cd_step(P, G, v0, y0, 1, 7, step[0] * 4, 0, None, None, True, 0.05, 1.0, float(B), 0.0, ws)
            step[0] += 1
        res[f"mode{mode}_us"] = round(timeit(f), 2)
        ops.check_chain_status()
    print(json.dumps({"shape": [V, H, C, B], **res}), flush=True)
