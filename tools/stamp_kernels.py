#!/usr/bin/env python
"""tools/stamp_kernels.py -- where the cycles of the REAL strip kernels go (diagnostic; GPU box).

Builds csrc/rbm_kernels.hip with -DQBM_DIAG_STAMPS into tools/microbench/bin/libqbm_stamp.so (never into the product
library), launches k_hidden / k_visible / k_grad at the benchmark shape (784 x 512, B = 256, C = 10, uint8 states)
and prints, per kernel, the median / p10 / p90 over waves of the s_memtime differences between the stamp slots:

  10 kernel entry   11 before the contraction (prologue done)   0 strip_mma entry   1 first LDS-DMA issued
  2 first quad landed   3 MFMA loop done   4 partial strips combined   12 kernel exit
  (k_grad: the second contraction uses slots 5..9)

    python tools/stamp_kernels.py [--build-only]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "qbm_image_classification_amd", "csrc")
SO = os.path.join(ROOT, "tools", "microbench", "bin", "libqbm_stamp.so")


def build(extra=(), so=None):
    from qbm_image_classification_amd.build import FLAGS

    so = so or SO
    os.makedirs(os.path.dirname(so), exist_ok=True)
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-DQBM_DIAG_STAMPS", *extra, os.path.join(CSRC, "rbm_kernels.hip"), "-o", so]
    subprocess.check_call(cmd)
    return so


def main():
    global SO
    variant = [a for a in sys.argv[1:] if a.startswith("--variant=")]
    if variant:      # --variant=name:-DFLAG=1,-DOTHER=2  -> a stamped library of its own
        name, _, flags = variant[0][len("--variant="):].partition(":")
        SO = SO.replace(".so", f"_{name}.so")
        if not os.path.exists(SO) or "--build-only" in sys.argv:
            build([f for f in flags.split(",") if f], SO)
    elif not os.path.exists(SO) or "--build-only" in sys.argv:
        build()
    if "--build-only" in sys.argv:
        return
    import numpy as np
    import torch

    from qbm_image_classification_amd import _lib

    lib = C.CDLL(SO)
    for name, argtypes in _lib.SIGNATURES.items():
        if name.startswith("qbm_rbm_"):       # this library holds the rbm_kernels.hip translation unit only
            getattr(lib, name).argtypes = argtypes
            getattr(lib, name).restype = C.c_int
    lib.qbm_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
    V, H, NC, B = 784, 512, 10, int(os.environ.get("QBM_STAMP_B", 256))
    dev = torch.device("cuda:0")
    torch.manual_seed(42)
    t = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    wld = int(os.environ.get("QBM_HACK_WLD", H))      # timing-only experiment: padded row pitch of W
    W = (torch.randn(V, wld) * 0.1).to(dev)
    b, d, U, c = torch.full((V,), 0.5, device=dev), torch.zeros(H, device=dev), torch.zeros(NC, H, device=dev), torch.zeros(NC, device=dev)
    P = _lib.RbmParams(*(C.c_void_p(x.data_ptr()) for x in (W, b, d, U, c)), V, H, NC)
    v0 = (torch.rand(B, V, device=dev) < 0.5).to(torch.uint8)
    y0 = torch.nn.functional.one_hot(torch.randint(0, NC, (B,), device=dev), NC).float()
    ph0, phk, h, vk, yk = t(B, H), t(B, H), torch.empty(B, H, dtype=torch.uint8, device=dev), torch.empty(B, V, dtype=torch.uint8, device=dev), t(B, NC)
    gW, gb, gd, gU, gc = t(V, H), t(V), t(H), t(NC, H), t(NC)
    G = _lib.RbmGrads(None, *(C.c_void_p(x.data_ptr()) for x in (gb, gd, gU, gc)))
    Wsc = W.clone()
    p = lambda x: C.c_void_p(x.data_ptr())
    kernels = {
        "k_hidden": (lambda: lib.qbm_rbm_sample_hidden_ex(C.byref(P), p(v0), 1, p(y0), B, p(ph0), None, p(h), 42, 0, 0, None), 256 * 8),
        "k_visible": (lambda: lib.qbm_rbm_sample_visible_class_ex(C.byref(P), p(h), 1, B, None, None, p(vk), None, p(yk), 42, 1, 2, 0, None), (196 + 256) * 8),
        "k_grad": (lambda: lib.qbm_rbm_two_phase_grad(V, H, NC, B, p(v0), 1, p(y0), p(ph0), p(vk), 1, p(yk), p(phk), C.byref(G), p(Wsc), None,
                                                     0.05, 1.0, float(B), 0.0, None), 238 * 8),
    }
    sync = torch.zeros(_lib.QBM_CD_SYNC_WORDS, dtype=torch.int32, device=dev)
    cws = _lib.CdWorkspace(p(ph0), p(h), p(vk), p(yk), p(phk), p(sync))
    lib.qbm_rbm_fused_chain_mode(2)     # the fused kernel also for CD-1 (default: from k = 2 on)
    kernels["k_chain"] = (lambda: lib.qbm_rbm_cd_chain(C.byref(P), p(v0), 1, p(y0), B, 1, 42, 0, 0, None, None, C.byref(cws), None), 256 * 8)
    # the fused tail of a CD-1 step (k_hgrad, mode 3): last hidden phase + gradient + update in one launch
    lib.qbm_rbm_cd_tail.argtypes = _lib.SIGNATURES["qbm_rbm_cd_tail"]
    rnd = [0]

    def tail():
        rnd[0] += 1
        lib.qbm_rbm_fused_chain_mode(3)      # an opt-in
        try:
            return lib.qbm_rbm_cd_tail(C.byref(P2), C.byref(G), p(v0), 1, p(y0), B, p(vk), p(yk), C.byref(cws), 1, 0.05, 1.0, float(B), 0.0, rnd[0], None)
        finally:
            lib.qbm_rbm_fused_chain_mode(2)
    keep = [b.clone(), d.clone(), U.clone(), c.clone()]
    P2 = _lib.RbmParams(p(Wsc), *(p(x) for x in keep), V, H, NC)
    kernels["k_hgrad"] = (tail, 256 * 8)
    for f, _ in kernels.values():    # warm: code objects loaded, caches in their steady state
        for _ in range(3):
            assert f() == 0
    torch.cuda.synchronize()
    phk.copy_(ph0)
    seq = {"k_hidden": [(10, 11, "prologue (loads, class term, Philox)"), (11, 0, "to strip_mma entry"), (0, 1, "first LDS-DMA quads issued"),
                        (1, 2, "first quad landed"), (2, 3, "MFMA loop"), (3, 4, "combine partial strips"), (4, 12, "epilogue"), (10, 12, "WAVE LIFETIME")],
           "k_visible": [(10, 11, "prologue"), (0, 1, "first LDS-DMA quads issued"), (1, 2, "first quad landed"), (2, 3, "MFMA loop"),
                         (3, 4, "combine partial strips"), (4, 12, "epilogue"), (10, 12, "WAVE LIFETIME")],
           # k_grad: wavefronts 0..3 of a workgroup contract phase 0 (slots 0..4), wavefronts 4..7 phase 1 (slots 5..9)
           "k_grad": [(10, 11, "prologue"), (0, 1, "phase 0: LDS-DMA issued"), (1, 2, "phase 0: quad landed"), (2, 3, "phase 0: MFMA loop"),
                      (3, 4, "phase 0: combine"), (4, 12, "phase 0 group: swap + epilogue"), (5, 6, "phase 1: LDS-DMA issued"),
                      (6, 7, "phase 1: quad landed"), (7, 8, "phase 1: MFMA loop"), (8, 9, "phase 1: combine"),
                      (9, 12, "phase 1 group: swap + epilogue"), (10, 12, "WAVE LIFETIME")]}
    # fused chain (CD-1): slots 5 entry, 6/7 before/after the first barrier, 8/9 the second, 13 exit; the phases' own stamps
    # (0..4, 11, 12) are those of the LAST hidden phase
    seq["k_chain"] = [(5, 6, "hidden phase 0 (cold)"), (6, 7, "barrier 1 (incl. waiting for the slowest workgroup)"),
                      (7, 8, "visible phase + class rows"), (8, 9, "barrier 2"), (9, 13, "hidden phase 1"),
                      (11, 0, "  last hidden phase: to strip_mma entry"), (0, 1, "  requests issued"), (1, 2, "  first quad landed"),
                      (2, 3, "  MFMA loop"), (3, 4, "  combine"), (4, 12, "  epilogue"), (5, 13, "WAVE LIFETIME")]
    # fused tail: 5 entry, 6 / 7 before / after the barrier of the column group (the hidden phase's own slots are re-used
    # below), 8..12 the contraction of phase 0 of the workgroup's W item (its requests went out under the hidden phase),
    # 0..4 that of phase 1, 13 exit
    seq["k_hgrad"] = [(5, 6, "hidden phase (+ requests of the gradient's operands)"), (6, 7, "barrier (incl. waiting for the slowest workgroup)"),
                      (7, 8, "to the contraction of phase 0 (group 0: requests of p(h_k))"), (8, 10, "  wait for the operands"), (10, 11, "  MFMA loop"),
                      (11, 12, "  combine"), (12, 0, "to the contraction of phase 1"), (0, 2, "  wait for the operands"), (2, 3, "  MFMA loop"),
                      (3, 4, "  combine"), (4, 13, "epilogue + other items"), (7, 13, "GRADIENT PART"), (5, 13, "WAVE LIFETIME")]
    for name, (f, _) in kernels.items():     # back-to-back launch time of the stamped build (stamps cost a few %)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            f()
        e1.record()
        e1.synchronize()
        print(f"{name}: {e0.elapsed_time(e1) / 300 * 1e3:.2f} us per launch (stamped build, back to back)")
    for name, (f, n_waves) in kernels.items():
        res = []
        for rep in range(5):
            lib.qbm_diag_clear_stamps()
            assert f() == 0
            torch.cuda.synchronize()
            buf = np.zeros((n_waves, 16), np.uint64)
            lib.qbm_diag_read_stamps(buf.ctypes.data_as(C.c_void_p), n_waves)
            res.append(buf)
        st = res[-1].astype(np.int64)
        ok = ((st[:, 10] != 0) & (st[:, 12] != 0) & (st[:, 11] != 0)) if name not in ("k_chain", "k_hgrad") else (st[:, 13] != 0)     # workgroups that ran to the end
        st = st[ok]
        print(f"== {name}: {ok.sum()} stamped waves (s_memtime is per XCD: only differences inside one wave are meaningful)")
        for a, bb, what in seq[name]:
            sel = (st[:, a] != 0) & (st[:, bb] != 0)
            if not sel.any():
                continue
            dts = (st[sel, bb] - st[sel, a]).astype(np.float64)
            print(f"   {what:40s} median {np.median(dts):8.0f}   p10 {np.percentile(dts, 10):8.0f}   p90 {np.percentile(dts, 90):8.0f}   ({sel.sum()} waves)")


if __name__ == "__main__":
    main()
