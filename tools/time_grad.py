"""k_grad with and without the bias / class-weight workgroups (which kinds are the long pole?)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qbm_image_classification_amd import _lib
from qbm_image_classification_amd._lib import RbmGrads, ptr, current_stream, check, QBM_U8
from qbm_image_classification_amd.rbm_ops import HipRbmOps

ops = HipRbmOps("cuda:0"); lib = ops.lib
V, H, Cn, B = 784, 512, 10, 256
d = ops.device
A0 = (torch.rand(B, V, device=d) < 0.5).to(torch.uint8); A1 = (torch.rand(B, V, device=d) < 0.5).to(torch.uint8)
Y0 = torch.nn.functional.one_hot(torch.randint(0, Cn, (B,), device=d), Cn).float(); Y1 = Y0.roll(1, 0).contiguous()
P0 = torch.rand(B, H, device=d); P1 = torch.rand(B, H, device=d)
gW = torch.empty(V, H, device=d); gb = torch.empty(V, device=d); gd = torch.empty(H, device=d); gU = torch.empty(Cn, H, device=d); gc = torch.empty(Cn, device=d)

def run(gs, reps=300):
    f = lambda: check(lib.qbm_rbm_two_phase_grad(V, H, Cn, B, ptr(A0), QBM_U8, ptr(Y0), ptr(P0), ptr(A1), QBM_U8, ptr(Y1), ptr(P1),
                                                 C.byref(gs), None, None, 0.0, 0.0, 1.0, 0.0, current_stream()), "grad")
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps // 20): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (reps // 20 * 20) * 1e3

print("all kinds        %.2f us" % run(RbmGrads(ptr(gW), ptr(gb), ptr(gd), ptr(gU), ptr(gc))))
print("no gb, gc        %.2f us" % run(RbmGrads(ptr(gW), None, ptr(gd), ptr(gU), None)))
print(RbmGrads._fields_)
