"""k_linear_fields at the Disc_QBM benchmark shape (B = 73, D = 784, H = 256, both phases), back to back."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from qbm_image_classification_amd.sampler import HipQuboOps
q = HipQuboOps("cuda:0")
rs = np.random.RandomState(0)
B, D, H = 73, 784, 256
X = torch.from_numpy((rs.random_sample((B, D)) < 0.13).astype(np.float64)).cuda()
Y = torch.from_numpy(rs.randint(0, 2, B).astype(np.int32)).cuda()
theta = torch.from_numpy(rs.uniform(-1, 1, (1 + D) * H + D + H + 1)).cuda()
for lab, name in ((Y, "clamped"), (None, "unclamped")):
    for _ in range(5): q.linear_fields_dev(X, lab, theta, D, H)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): q.linear_fields_dev(X, lab, theta, D, H)
    torch.cuda.synchronize()
    print(json.dumps({"what": f"k_linear_fields {name} B=73 D=784 H=256, back to back (eager launches)", "us_per_launch": round((time.perf_counter() - t0) / 200 * 1e6, 2)}))
