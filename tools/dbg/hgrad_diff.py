import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from helpers import synth_inputs, synth_params
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet

ops = HipRbmOps("cuda:0")
V, H, C, B, k = 784, 512, 10, 256, 1
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
W, b, d, U, c = synth_params(V, H, C, 400 + V + H)
v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 10)
res = {}
for mode in (0, 1):
    ops.fused_chain_mode(mode)
    for fuse in (False, True):
        P = ParamSet(t(W), t(b), t(d), t(U), t(c))
        G = GradSet(V, H, C, ops.device)
        ws = ops.cd_workspace(P, B)
        out = ops.cd_step(P, G, t(v_bin.astype(np.uint8)), t(y1h), k, 99, 1000, 512, None, None, fuse, 0.05, 1.0, float(B), 0.0, ws)
        torch.cuda.synchronize()
        res[(mode, fuse)] = dict(W=P.W.cpu().numpy(), gW=G.gW.cpu().numpy(), phk=out[4].cpu().numpy(), ph0=out[0].cpu().numpy(),
                                 st=ws[5][512:516].tolist(), d=P.d.cpu().numpy(), b=P.b.cpu().numpy(), gb=G.gb.cpu().numpy())
for fuse in (False, True):
    a, bb = res[(0, fuse)], res[(1, fuse)]
    print("fuse", fuse, "status", bb["st"])
    for nm in ("phk", "ph0", "gW", "W", "d", "b", "gb"):
        x, y = a[nm], bb[nm]
        bad = x.view(np.uint32) != y.view(np.uint32)
        if bad.ndim == 2:
            rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
            print(f"  {nm}: {bad.sum()} differ; rows {rows.min() if rows.size else None}..{rows.max() if rows.size else None} ({rows.size}), cols {cols.min() if cols.size else None}..{cols.max() if cols.size else None} ({cols.size})")
            if bad.any() and nm in ("gW", "W"):
                # block structure 128 x 16
                blk = bad.reshape(-1, 16, H // 16, 16) if x.shape[0] % 16 == 0 else None
                tiles = bad[:784 // 16 * 16].reshape(49, 16, 32, 16).any(axis=(1, 3))
                print("   tiles (49 row tiles x 32 col groups), X = differs:")
                for r in range(49):
                    print("   ", "".join("X" if v else "." for v in tiles[r]))
                i, j = np.argwhere(bad)[0]
                print("   first bad", i, j, x[i, j], y[i, j])
        else:
            print(f"  {nm}: {bad.sum()} differ")
