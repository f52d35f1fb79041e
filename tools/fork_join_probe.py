"""Does a fork / join inside a captured hipGraph buy overlap on this stack?  (branch A = k_visible ->
k_hidden, branch B = the positive half of the gradient, join -> the rest.)  Probe with the kernels that exist: a CD-1 step's
four launches on the main branch, and an INDEPENDENT stand-in of about the size of a phase-0 gradient launch (k_free_energy on
other buffers: ~6-8 us, no dynamic LDS, so it can share CUs with every strip kernel) either (a) in sequence behind the step,
(b) forked off behind the first launch and joined in front of the last.  One JSON line: us per step of the plain step, of (a)
and of (b).  (b) - plain = what the fork / join itself costs when the overlap is perfect; (a) - (b) = what overlap buys."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from qbm_image_classification_amd._lib import QBM_U8, check, ptr
from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet

ops = HipRbmOps("cuda:0"); lib = ops.lib; dev = ops.device
V, H, NC, B = 784, 512, 10, 256
torch.manual_seed(42)
P = ParamSet((torch.randn(V, H) * 0.1).to(dev), torch.zeros(V, device=dev), torch.zeros(H, device=dev), torch.zeros(NC, H, device=dev), torch.zeros(NC, device=dev))
G = GradSet(V, H, NC, dev)
v0 = (ops.uniform(1, 0, 0, B, V) < 0.5).to(torch.uint8)
y0 = torch.nn.functional.one_hot(torch.randint(0, NC, (B,), device=dev), NC).float()
ph0, phk = torch.empty(B, H, device=dev), torch.empty(B, H, device=dev)
h, vk, yk = torch.empty(B, H, dtype=torch.uint8, device=dev), torch.empty(B, V, dtype=torch.uint8, device=dev), torch.empty(B, NC, device=dev)
x = torch.rand(B, V, device=dev); pre = torch.empty(B, H, device=dev); nf = torch.empty(B, NC, device=dev); pp = torch.empty(B, NC, device=dev)
ps, gs = P.struct(), G.struct(with_gW=False)
pre.copy_(torch.randn(B, H, device=dev))
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def k1(): check(lib.qbm_rbm_sample_hidden_ex(C.byref(ps), ptr(v0), QBM_U8, ptr(y0), B, ptr(ph0), None, ptr(h), 42, 0, 0, S()), "h1")
def k2(): check(lib.qbm_rbm_sample_visible_class_ex(C.byref(ps), ptr(h), QBM_U8, B, None, None, ptr(vk), None, ptr(yk), 42, 1, 2, 0, S()), "v")
def k3(): check(lib.qbm_rbm_sample_hidden_ex(C.byref(ps), ptr(vk), QBM_U8, ptr(yk), B, ptr(phk), None, None, 42, 3, 0, S()), "h2")
def k4(): check(lib.qbm_rbm_two_phase_grad(V, H, NC, B, ptr(v0), QBM_U8, ptr(y0), ptr(ph0), ptr(vk), QBM_U8, ptr(yk), ptr(phk), C.byref(gs), ptr(P.W), None, 1e-9, 1.0, float(B), 0.0, S()), "g")
def side():   # stand-in: the free-energy rows of an unrelated buffer (x.W + d is NOT recomputed: only the row kernel)
    check(lib.qbm_rbm_free_energy(C.byref(ps), ptr(x), B, ptr(pre), ptr(nf), ptr(pp), S()), "fe")

def plain():
    k1(); k2(); k3(); k4()
def seq():
    k1(); k2(); k3(); side(); k4()
side_stream = torch.cuda.Stream()
def forked():
    k1()
    main = torch.cuda.current_stream()
    e = torch.cuda.Event(); e.record(main)
    side_stream.wait_event(e)
    with torch.cuda.stream(side_stream):
        side()
        e2 = torch.cuda.Event(); e2.record(side_stream)
    k2(); k3()
    main.wait_event(e2)
    k4()

def time(fn, steps=8, reps=200):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(steps): fn()
    for _ in range(10): g.replay()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): g.replay()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * steps) * 1e3)
    return round(best, 2)
# note: qbm_rbm_free_energy = k_hidden<MODE 1> + k_free_energy: two launches (~13 us) as the stand-in
print(json.dumps({"plain_step_us": time(plain), "step_plus_standin_in_sequence_us": time(seq), "step_with_standin_forked_us": time(forked)}))
