"""CD-1 step time (hipGraph of 8 steps, 200 replays) with the fused tail (k_hgrad) and with one launch per phase."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet

ops = HipRbmOps("cuda:0")
dev = ops.device
MODE_TAIL = 3
for V, H, C, B in ((784, 128, 2, 128), (784, 512, 10, 256), (784, 512, 10, 128), (784, 256, 10, 256), (784, 512, 10, 64), (784, 128, 10, 256)):
    res = {}
    for mode in (0, MODE_TAIL):
        ops.fused_chain_mode(mode)
        torch.manual_seed(1)
        P = ParamSet((torch.randn(V, H) * 0.1).to(dev), torch.zeros(V, device=dev), torch.zeros(H, device=dev), torch.zeros(C, H, device=dev), torch.zeros(C, device=dev))
        G = GradSet(V, H, C, dev)
        v0 = (ops.uniform(1, 0, 0, B, V) < 0.5).to(torch.uint8)
        y0 = torch.nn.functional.one_hot(torch.randint(0, C, (B,), device=dev), C).float()
        ws = ops.cd_workspace(P, B)
        step = lambda i: ops.cd_step(P, G, v0, y0, 1, 7, 4 * i, 0, None, None, True, 0.05, 1.0, float(B), 0.0, ws)
        for i in range(8):
            step(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(8):
                step(i)
        for _ in range(20):
            g.replay()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                g.replay()
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 1600 * 1e3)
        assert int(ws[5][512]) == 0
        res[mode] = round(best, 2)
    print(json.dumps({"shape": [V, H, C, B], "us_per_step_launch_per_phase": res[0], "us_per_step_fused_tail": res[MODE_TAIL]}), flush=True)
