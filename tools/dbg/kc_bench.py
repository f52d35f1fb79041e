import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from qbm_image_classification_amd import coreset
rs = np.random.RandomState(0)
emb = torch.from_numpy(rs.standard_normal((12665, 256)).astype(np.float32)).cuda()
coreset.kcenter_picks(emb, [0], 1001, fused=True)
dt = 1e9
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); coreset.kcenter_picks(emb, [0], 1001, fused=True); torch.cuda.synchronize(); dt = min(dt, time.perf_counter() - t0)
print(json.dumps({"what": "k_center_greedy 12665 x 256, 1000 centres, one launch", "us_per_pick": round(dt / 1001 * 1e6, 3)}))
