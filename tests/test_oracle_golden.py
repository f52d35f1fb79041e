"""CPU: pin the oracle (oracle/qbm_oracle.c) against golden vectors produced by the imported
reference (tools/make_golden.py).  Tolerances are those of SURVEY.md section 8.D:
probabilities <= 1e-6 abs, -F <= 1e-5 relative, identical argmax, parameters after three
discriminative steps <= 1e-5 relative (on elements that are not ~0)."""
import numpy as np
import pytest

from helpers import golden, rel_err, synth_inputs, synth_params

CFGS = ["small", "mid", "big"]
# p(y|x) = softmax of fp32 free energies with |F| up to ~550: the reference's own fp32 result sits
# 1.4e-4 from an fp64 evaluation ("Free-energy tolerance"); argmax must be equal.
P_TOL = 2e-4


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ([0, 0], [0, 0, 0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 2, [0xFFFFFFFF] * 4, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0xA4093822, 0x299F31D0], [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for key, ctr, want in kat:
        assert oracle.philox_raw(key, ctr).tolist() == want


def test_uniform_layout(oracle):
    u = oracle.philox_uniform(seed=(7 << 32) | 5, offset=(3 << 32) | 9, row0=11, rows=3, cols=10)
    for i in range(3):
        for j in range(10):
            w = oracle.philox_raw([5, 7], [j >> 2, 11 + i, 9, 3])[j & 3]
            assert u[i, j] == np.float32(w >> 8) * np.float32(2.0 ** -24)
    assert (u >= 0).all() and (u < 1).all()


def test_det_math_accuracy(oracle):
    x = np.linspace(-87, 88.7, 100001).astype(np.float32)
    assert rel_err(oracle.math("exp", x), np.exp(x.astype(np.float64))) < 2.5e-7
    x = np.exp(np.linspace(-17, 80, 100001)).astype(np.float32)
    ref = np.log(x.astype(np.float64))
    assert np.max(np.abs(oracle.math("log", x) - ref) / np.maximum(np.abs(ref), 1.0)) < 2.5e-7
    assert oracle.math("exp", np.array([89.0, -200.0])).tolist() == [np.inf, 0.0]
    assert oracle.math("log", np.array([0.0, np.inf])).tolist() == [-np.inf, np.inf]
    # unstabilised softplus overflows like the reference's log(1 + exp(z)) (ClassificationRBM.py:72)
    assert np.isinf(oracle.math("softplus", np.array([100.0]))[0])


@pytest.mark.parametrize("nq,perm", [(4, False), (8, True), (8, False)])
def test_qgemm_is_split_fma_chain(oracle, nq, perm):
    """[Q]: nq contiguous chains of roundup16(ceil(K/nq)) terms, each an fmaf chain, summed in chain order (8 chains: the
    conditionals; 4: the gradient blocks).  [Q']: the conditionals visit every aligned 16-term piece in the order
    0,4,8,12, 1,5,9,13, 2,6,10,14, 3,7,11,15 (terms beyond K skipped); the gradient in ascending order."""
    order = [4 * (j & 3) + (j >> 2) for j in range(16)] if perm else list(range(16))
    rs = np.random.RandomState(0)
    for (M, N, K) in [(3, 5, 16), (2, 4, 9), (4, 3, 784), (1, 2, 1), (2, 2, 5), (2, 3, 100), (2, 2, 512)]:
        A = rs.standard_normal((M, K)).astype(np.float32)
        B = rs.standard_normal((K, N)).astype(np.float32)
        got = oracle.qgemm(A, B, nq, perm)
        Kq = ((K + nq - 1) // nq + 15) // 16 * 16
        want = np.zeros((M, N), np.float32)
        for m in range(M):
            for n in range(N):
                parts = []
                for q in range(nq):
                    acc = np.float32(0)
                    k1 = min(K, (q + 1) * Kq)
                    for k in (p0 + j for p0 in range(q * Kq, k1, 16) for j in order if p0 + j < k1):
                        # exact fma in float64: a*b is exact (48 bits), the sum is rounded once to
                        # 53 bits and again to 24 -- double rounding can differ only on ties
                        acc = np.float32(np.float64(A[m, k]) * np.float64(B[k, n]) + np.float64(acc))
                    parts.append(acc)
                t = parts[0]
                for q in range(1, nq):
                    t = np.float32(t + parts[q])
                want[m, n] = t
        assert np.max(np.abs(got - want)) <= 1e-6 * max(1.0, np.abs(want).max())
        assert np.allclose(got, A.astype(np.float64) @ B.astype(np.float64), rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", CFGS)
def test_conditionals_match_reference(oracle, name):
    g = golden(f"rbm_prims_{name}.npz")
    V, H, C, B, seed = (int(g[k]) for k in ("V", "H", "C", "B", "seed"))
    W, b, d, U, c = synth_params(V, H, C, seed)
    v_bin, x_grey, h_bin, labels, y1h = synth_inputs(B, V, H, C, seed)
    assert np.array_equal(v_bin.astype(np.uint8), g["v_bin"]) and np.array_equal(x_grey, g["x_grey"])
    ph, _ = oracle.rbm_sample_hidden(v_bin, y1h, W, d, U, sample=False)
    assert np.max(np.abs(ph - g["ph_bin"])) <= 1e-6
    ph, _ = oracle.rbm_sample_hidden(x_grey, y1h, W, d, U, sample=False)
    assert np.max(np.abs(ph - g["ph_grey"])) <= 1e-6
    pv, _ = oracle.rbm_sample_visible(h_bin, W, b, sample=False)
    assert np.max(np.abs(pv - g["pv"])) <= 1e-6
    py, _ = oracle.rbm_sample_class(h_bin, U, c, sample=False)
    assert np.max(np.abs(py - g["py"])) <= 1e-6


@pytest.mark.parametrize("name", CFGS)
def test_free_energy_matches_reference(oracle, name):
    g = golden(f"rbm_prims_{name}.npz")
    V, H, C, B, seed = (int(g[k]) for k in ("V", "H", "C", "B", "seed"))
    W, b, d, U, c = synth_params(V, H, C, seed)
    v_bin, x_grey, *_ = synth_inputs(B, V, H, C, seed)
    for x, tag in ((x_grey, "grey"), (v_bin, "bin")):
        pre, nf, p = oracle.rbm_free_energy(x, W, d, U, c)
        assert rel_err(nf, g[f"negF_{tag}"]) <= 1e-5          # vs the reference's own fp32 loop
        assert rel_err(nf, g[f"negF_{tag}_f64"]) <= 1e-5      # vs the fp64 companion
        assert np.max(np.abs(p - g[f"pyx_{tag}"])) <= P_TOL
        assert np.array_equal(p.argmax(1), g[f"pyx_{tag}"].argmax(1))


@pytest.mark.parametrize("name", CFGS)
def test_discriminative_steps_match_reference(oracle, name):
    g = golden(f"rbm_disc_{name}.npz")
    V, H, C, B, seed = (int(g[k]) for k in ("V", "H", "C", "B", "seed"))
    lr, sparse = float(g["lr"]), float(g["sparse"])
    sr, sc = (int(t) for t in g["sub"])
    W, b, d, U, c = synth_params(V, H, C, seed)
    x, labels = g["x"], g["labels"]
    for step in range(3):
        xs = np.roll(x, step, axis=0).copy()
        ls = np.roll(labels, step).copy()
        gW, gd, gU, gc, p, nf = oracle.rbm_disc_grad(xs, ls, W, d, U, c)
        if step == 0:
            scale = np.abs(g["gW0"]).max()
            assert np.max(np.abs(gW[::sr, ::sc] - g["gW0"])) <= 1e-5 * scale
            assert np.max(np.abs(gU - g["gU0"])) <= 1e-5 * max(1.0, np.abs(g["gU0"]).max())
            assert np.max(np.abs(gd - g["gd0"])) <= 1e-5 * max(1.0, np.abs(g["gd0"]).max())
            assert np.max(np.abs(gc - g["gc0"])) <= 1e-5 * max(1.0, np.abs(g["gc0"]).max())
        assert np.max(np.abs(p - g[f"p{step}"])) <= P_TOL
        assert np.array_equal(p.argmax(1), g[f"pred{step}"])
        oracle.rbm_update(W, b, d, U, c, gW, None, gd, gU, gc, lr, 1.0, float(B), sparse)
    assert np.max(np.abs(W[::sr, ::sc] - g["W_after"])) <= 1e-5 * np.abs(g["W_after"]).max()
    assert np.max(np.abs(b - g["b_after"])) <= 1e-6
    assert np.max(np.abs(d - g["d_after"])) <= 1e-5
    assert np.max(np.abs(U - g["U_after"])) <= 1e-5
    assert np.max(np.abs(c - g["c_after"])) <= 1e-5


def test_torch_cpu_baseline_conditionals_match_reference_golden():
    """bench.py's cpu_baseline leg (oracle/rbm_torch_cpu.py) computes the reference's conditionals: checked against the
    imported reference's outputs (<= 1e-6), and one CD-1 step moves the parameters the right way."""
    import torch

    from oracle.rbm_torch_cpu import TorchCpuRbm

    g = golden("rbm_prims_mid.npz")
    V, H, C, B, seed = (int(g[k]) for k in ("V", "H", "C", "B", "seed"))
    W, b, d, U, c = synth_params(V, H, C, seed)
    v_bin, x_grey, h_bin, labels, y1h = synth_inputs(B, V, H, C, seed)
    r = TorchCpuRbm(W, b, d, U, c, lr=0.05, threads=2)
    t = torch.from_numpy
    assert np.max(np.abs(r.p_hidden(t(v_bin), t(y1h)).numpy() - g["ph_bin"])) <= 1e-6
    assert np.max(np.abs(r.p_visible(t(h_bin)).numpy() - g["pv"])) <= 1e-6
    assert np.max(np.abs(r.p_class(t(h_bin)).numpy() - g["py"])) <= 1e-6
    torch.manual_seed(0)
    v1, y1, ph0, ph1 = r.cd1_step(t(v_bin), t(y1h))
    assert set(np.unique(v1.numpy())) <= {0.0, 1.0} and (y1.sum(1) == 1).all()
    want = W + 0.05 / B * (v_bin.T @ ph0.numpy() - v1.numpy().T @ ph1.numpy())
    assert np.max(np.abs(r.W.numpy() - want)) <= 1e-5
