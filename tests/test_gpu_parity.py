"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs.  Bar: bit-exact for sampled bits AND for every fp32 value
(the numerical contract makes the two sides perform the same IEEE operations in the same order);
against the reference's golden vectors the tolerances of SURVEY.md 8.D apply."""
import numpy as np
import pytest

from helpers import golden, rel_err, synth_inputs, synth_params

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SHAPES = [  # (V, H, C, B)
    (16, 8, 2, 6),        # Bars-and-Stripes size (config #1)
    (13, 9, 3, 5),        # nothing aligned: scalar guarded paths
    (784, 9, 2, 7),       # rbm_main.py defaults (H=9, batch 7)
    (784, 128, 2, 128),   # config #2
    (784, 512, 10, 256),  # config #3/#4 per-GPU shape
    (784, 512, 2, 33),    # ragged last strip
    (100, 260, 4, 1),     # single row
    (144, 256, 3, 70),    # 16 column strips of sample_hidden: dealt to the XCDs in pairs (H a multiple of 256), 2 grid planes
    (96, 768, 2, 40),     # ... 48 strips, 6 grid planes
]


@pytest.fixture(scope="module")
def ops():
    from qbm_image_classification_amd.rbm_ops import HipRbmOps

    return HipRbmOps("cuda:0")  # raises if the HIP library or the GPU is missing


def P_of(ops, W, b, d, U, c):
    from qbm_image_classification_amd.rbm_ops import ParamSet

    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ops.device)
    return ParamSet(t(W), t(b), t(d), t(U), t(c))


def dev(ops, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(ops.device)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bitwise(got, want, what):
    got = got.detach().cpu().numpy() if hasattr(got, "detach") else np.asarray(got)
    bad = bits(got) != bits(want)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} elements differ, max abs {np.abs(got - want).max()}"


def test_native_library_is_loaded(ops):
    import os

    with open("/proc/self/maps") as f:
        assert "libqbm_hip.so" in f.read()
    assert os.path.exists(ops.lib._name)


def test_device_math_bitwise(ops, oracle):
    xs = np.concatenate([np.linspace(-100, 100, 200001), [0.0, -0.0, 88.72, 88.73, -87.0, -87.01, 1e-30, np.inf, -np.inf]])
    xs = xs.astype(np.float32)
    for which in ("exp", "sigmoid", "softplus"):
        assert_bitwise(ops.math(which, dev(ops, xs)), oracle.math(which, xs), which)
    xl = np.concatenate([np.exp(np.linspace(-40, 88, 200001)), [0.0, 1.0, np.inf, 2.0 ** -24, 1 - 2.0 ** -24]]).astype(np.float32)
    assert_bitwise(ops.math("log", dev(ops, xl)), oracle.math("log", xl), "log")


def test_philox_uniform_bitwise(ops, oracle):
    for seed, off, row0, r, c in [(0, 0, 0, 4, 8), ((7 << 32) | 5, (3 << 32) | 9, 11, 33, 130), (2 ** 64 - 1, 2 ** 63, 2 ** 32 - 2, 3, 7)]:
        assert_bitwise(ops.uniform(seed, off, row0, r, c), oracle.philox_uniform(seed, off, row0, r, c), "uniform")


@pytest.mark.parametrize("V,H,C,B", SHAPES)
def test_conditionals_bitwise(ops, oracle, V, H, C, B):
    W, b, d, U, c = synth_params(V, H, C, 100 + V + H)
    v_bin, x_grey, h_bin, labels, y1h = synth_inputs(B, V, H, C, 7)
    P = P_of(ops, W, b, d, U, c)
    seed, off, row0 = 0x1234567887654321, 42, 1000
    for v in (v_bin, x_grey):
        p, s = ops.sample_hidden(P, dev(ops, v), dev(ops, y1h), seed, off, row0, True, True)
        po, so = oracle.rbm_sample_hidden(v, y1h, W, d, U, seed, off, row0)
        assert_bitwise(p, po, "p(h|v,y)")
        assert_bitwise(s, so, "h sample")
    p, _ = ops.sample_hidden(P, dev(ops, v_bin), None)
    assert_bitwise(p, oracle.rbm_sample_hidden(v_bin, None, W, d, U, sample=False)[0], "p(h|v)")
    assert_bitwise(ops.hidden_preact(P, dev(ops, x_grey)), oracle.rbm_free_energy(x_grey, W, d, U, c)[0], "x.W + d")
    p, s = ops.sample_visible(P, dev(ops, h_bin), seed, off + 1, row0, True, True)
    po, so = oracle.rbm_sample_visible(h_bin, W, b, seed, off + 1, row0)
    assert_bitwise(p, po, "p(v|h)")
    assert_bitwise(s, so, "v sample")
    p, s = ops.sample_class(P, dev(ops, h_bin), seed, off + 2, row0, True, True)
    po, so = oracle.rbm_sample_class(h_bin, U, c, seed, off + 2, row0)
    assert_bitwise(p, po, "p(y|h)")
    assert_bitwise(s, so, "y sample")
    assert (s.sum(1) == 1).all()
    # uint8 {0,1} states in, fp32 + uint8 samples out: identical bits to the fp32 path
    ph_o, sh_o = oracle.rbm_sample_hidden(v_bin, y1h, W, d, U, seed, off, row0)
    p8, sf, s8 = ops.sample_hidden_ex(P, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), seed, off, row0, True, True, True)
    assert_bitwise(p8, ph_o, "p(h|v u8,y)")
    assert_bitwise(sf, sh_o, "h sample (u8 input)")
    assert np.array_equal(s8.cpu().numpy(), sh_o.astype(np.uint8))
    pv_o, sv_o = oracle.rbm_sample_visible(h_bin, W, b, seed, off + 1, row0)
    pv, vf, v8, py, ys = ops.sample_visible_class(P, dev(ops, h_bin.astype(np.uint8)), seed, off + 1, off + 2, row0,
                                                  True, True, True, True, True)
    assert_bitwise(pv, pv_o, "p(v|h u8)")
    assert_bitwise(vf, sv_o, "v sample (u8 input)")
    assert np.array_equal(v8.cpu().numpy(), sv_o.astype(np.uint8))
    assert_bitwise(py, po, "p(y|h u8) riding along")
    assert_bitwise(ys, so, "y sample riding along")


@pytest.mark.parametrize("V,H,C,B", [(784, 512, 10, 544), (784, 512, 10, 2048), (256, 64, 2, 1200), (784, 512, 0, 1024)])
def test_large_batch_conditionals_bitwise(ops, oracle, V, H, C, B):
    """Minibatches with more strips than CUs (B > 256 at H = 512): sample_hidden switches to panels of 64 k (two workgroups
    per CU, two panels per chain at V = 784); uint8 states; C = 0 here: without the class term.  Same bits as the oracle."""
    with_y = C > 0
    C = C or 2
    W, b, d, U, c = synth_params(V, H, C, 700 + V + H)
    v_bin, _, h_bin, labels, y1h = synth_inputs(B, V, H, C, 17)
    P = P_of(ops, W, b, d, U, c)
    seed, off, row0 = 31, 9, 4096
    ph_o, sh_o = oracle.rbm_sample_hidden(v_bin, y1h if with_y else None, W, d, U, seed, off, row0)
    p8, sf, s8 = ops.sample_hidden_ex(P, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h) if with_y else None, seed, off, row0,
                                      True, True, True)
    assert_bitwise(p8, ph_o, "p(h|v u8,y)")
    assert np.array_equal(s8.cpu().numpy(), sh_o.astype(np.uint8))
    pv_o, sv_o = oracle.rbm_sample_visible(h_bin, W, b, seed, off + 1, row0)
    po, so = oracle.rbm_sample_class(h_bin, U, c, seed, off + 2, row0)
    pv, vf, v8, py, ys = ops.sample_visible_class(P, dev(ops, h_bin.astype(np.uint8)), seed, off + 1, off + 2, row0,
                                                  True, True, True, True, True)
    assert_bitwise(pv, pv_o, "p(v|h u8)")
    assert np.array_equal(v8.cpu().numpy(), sv_o.astype(np.uint8))
    assert_bitwise(py, po, "p(y|h u8)")
    assert_bitwise(ys, so, "y sample")


@pytest.mark.parametrize("V,H,C,B", SHAPES)
def test_free_energy_bitwise(ops, oracle, V, H, C, B):
    W, b, d, U, c = synth_params(V, H, C, 200 + V + H)
    _, x_grey, *_ = synth_inputs(B, V, H, C, 8)
    nf, p = ops.free_energy(P_of(ops, W, b, d, U, c), dev(ops, x_grey))
    _, nfo, po = oracle.rbm_free_energy(x_grey, W, d, U, c)
    assert_bitwise(nf, nfo, "-F")
    assert_bitwise(p, po, "p(y|x)")


@pytest.mark.parametrize("name", ["small", "mid", "big"])
def test_against_reference_golden(ops, name):
    """HIP vs the imported reference's outputs (tools/make_golden.py), SURVEY.md 8.D tolerances."""
    g = golden(f"rbm_prims_{name}.npz")
    V, H, C, B, seed = (int(g[k]) for k in ("V", "H", "C", "B", "seed"))
    W, b, d, U, c = synth_params(V, H, C, seed)
    v_bin, x_grey, h_bin, labels, y1h = synth_inputs(B, V, H, C, seed)
    P = P_of(ops, W, b, d, U, c)
    cpu = lambda t: t.cpu().numpy()
    assert np.max(np.abs(cpu(ops.sample_hidden(P, dev(ops, v_bin), dev(ops, y1h))[0]) - g["ph_bin"])) <= 1e-6
    assert np.max(np.abs(cpu(ops.sample_hidden(P, dev(ops, x_grey), dev(ops, y1h))[0]) - g["ph_grey"])) <= 1e-6
    assert np.max(np.abs(cpu(ops.sample_visible(P, dev(ops, h_bin))[0]) - g["pv"])) <= 1e-6
    assert np.max(np.abs(cpu(ops.sample_class(P, dev(ops, h_bin))[0]) - g["py"])) <= 1e-6
    for x, tag in ((x_grey, "grey"), (v_bin, "bin")):
        nf, p = ops.free_energy(P, dev(ops, x))
        assert rel_err(cpu(nf), g[f"negF_{tag}"]) <= 1e-5      # free energy: 1e-5 relative (north_star)
        assert rel_err(cpu(nf), g[f"negF_{tag}_f64"]) <= 1e-5
        assert np.array_equal(cpu(p).argmax(1), g[f"pyx_{tag}"].argmax(1))  # identical argmax labels
        assert np.max(np.abs(cpu(p) - g[f"pyx_{tag}"])) <= 2e-4


# + batch > one 256-row tile; + the benchmark shape at B = 1024: several panels per K chain of the gradient (uint8 inputs:
# the phase-parallel 512-thread kernel)
@pytest.mark.parametrize("V,H,C,B", SHAPES + [(100, 70, 15, 300), (64, 40, 5, 515), (784, 512, 10, 1024)])
def test_discriminative_step_bitwise(ops, oracle, V, H, C, B):
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 300 + V + H)
    v_bin, x_grey, _, labels, _ = synth_inputs(B, V, H, C, 9)
    # greyscale fp32 inputs (what the reference trains on), and binary inputs handed over as uint8 {0,1} (qbm_rbm_disc_step_ex:
    # the uint8 kernels) -- the oracle sees the same values as fp32
    for fuse, x_in, x_ref in ((False, dev(ops, x_grey), x_grey), (True, dev(ops, x_grey), x_grey),
                              (False, dev(ops, v_bin.astype(np.uint8)), v_bin), (True, dev(ops, v_bin.astype(np.uint8)), v_bin)):
        P = P_of(ops, W, b, d, U, c)
        G = GradSet(V, H, C, ops.device)
        sums = torch.tensor([1.5, -2.25], dtype=torch.float64, device=ops.device)   # the training loop's running loss sums
        p = ops.disc_step(P, G, x_in, torch.from_numpy(labels), True, fuse, 0.05, 1.0, float(B), 0.001, loss_acc=sums)
        gW, gd, gU, gc, po, _, rl, bl = oracle.rbm_disc_grad(x_ref, labels, W, d, U, c, with_loss=True)
        assert_bitwise(p, po, "p(y|x)")
        assert_bitwise(ops._disc_ws[1][2], rl, "per-row loss terms (CrossEntropyLoss on probabilities, -log(p_y + 1e-8))")
        assert sums.tolist() == [1.5 + float(bl[0]), -2.25 + float(bl[1])], "minibatch loss terms added to the float64 sums"
        if not fuse:
            assert_bitwise(G.gW, gW, "gW")
        assert_bitwise(G.gd, gd, "gd")
        assert_bitwise(G.gU, gU, "gU")
        assert_bitwise(G.gc, gc, "gc")
        Wn, bn, dn, Un, cn = (a.copy() for a in (W, b, d, U, c))
        oracle.rbm_update(Wn, bn, dn, Un, cn, gW, None, gd, gU, gc, 0.05, 1.0, float(B), 0.001)
        for got, want, nm in ((P.W, Wn, "W"), (P.b, bn, "b"), (P.d, dn, "d"), (P.U, Un, "U"), (P.c, cn, "c")):
            assert_bitwise(got, want, f"{nm} after update (fuse={fuse})")


@pytest.mark.parametrize("V,H,C,B,k,pcd", [(16, 8, 2, 6, 1, False), (13, 9, 3, 5, 2, True), (784, 128, 2, 128, 1, False),
                                           (784, 512, 10, 256, 1, False), (784, 512, 10, 64, 3, True),
                                           # more than one panel per K chain of the gradient (the panel is capped at 64 k, a
                                           # chain of B = 2048 rows is 512 k long): the multi-panel loop of the 512-thread
                                           # phase-parallel k_grad behind the B = 2048 numbers of DESIGN.md; B = 544: a
                                           # ragged last panel, persistent chains
                                           (784, 512, 10, 2048, 1, False), (784, 512, 10, 544, 2, True)])
def test_cd_step_bitwise(ops, oracle, V, H, C, B, k, pcd):
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 400 + V + H)
    v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 10)
    seed, off, row0 = 99, 1000, 512
    cv = (np.roll(v_bin, 1, axis=0).copy(), np.roll(y1h, 1, axis=0).copy()) if pcd else (None, None)
    ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, seed, off, row0,
                                           None if not pcd else cv[0], None if not pcd else cv[1])
    gW, gb, gd, gU, gc = oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
    # the visible data go in as fp32 (what the reference holds) or as uint8 {0,1}: same bits out; uint8 data take the
    # phase-parallel gradient kernel, with the update fused and with the gradients materialised (the data-parallel form)
    for fuse, as_u8 in ((False, False), (True, True), (False, True), (True, False)):
        P = P_of(ops, W, b, d, U, c)
        G = GradSet(V, H, C, ops.device)
        chain_v = dev(ops, np.roll(v_bin, 1, axis=0).astype(np.uint8)) if pcd else None
        chain_y = dev(ops, np.roll(y1h, 1, axis=0)) if pcd else None
        v0_in = dev(ops, v_bin.astype(np.uint8)) if as_u8 else dev(ops, v_bin)
        ws = ops.cd_step(P, G, v0_in, dev(ops, y1h), k, seed, off, row0, chain_v, chain_y, fuse, 0.05, 1.0,
                         float(B), 0.0)
        assert_bitwise(ws[0], ph0, "p(h0)")
        assert np.array_equal(ws[2].cpu().numpy(), vk.astype(np.uint8)), "v_k"
        assert_bitwise(ws[3], yk, "y_k")
        assert_bitwise(ws[4], phk, "p(h_k)")
        if pcd:
            assert np.array_equal(chain_v.cpu().numpy(), cv[0].astype(np.uint8)), "persistent v"
            assert_bitwise(chain_y, cv[1], "persistent y")
        if not fuse:
            assert_bitwise(G.gW, gW, "gW")
        for got, want, nm in ((G.gb, gb, "gb"), (G.gd, gd, "gd"), (G.gU, gU, "gU"), (G.gc, gc, "gc")):
            assert_bitwise(got, want, nm)
        if fuse:
            Wn, bn, dn, Un, cn = (a.copy() for a in (W, b, d, U, c))
            oracle.rbm_update(Wn, bn, dn, Un, cn, gW, gb, gd, gU, gc, 0.05, 1.0, float(B), 0.0)
            for got, want, nm in ((P.W, Wn, "W"), (P.b, bn, "b"), (P.d, dn, "d"), (P.U, Un, "U"), (P.c, cn, "c")):
                assert_bitwise(got, want, f"{nm} after fused update")


@pytest.mark.parametrize("V,H,C,B,k,mode", [(784, 512, 10, 256, 1, 1), (784, 512, 10, 256, 2, 2), (784, 128, 2, 128, 1, 0), (13, 9, 3, 5, 2, 1),
                                            (784, 512, 10, 544, 1, 1)])
def test_cd_step_monitor_bitwise(ops, oracle, V, H, C, B, k, mode):
    """The training loop's monitor riding in a CD-k step (qbm_rbm_cd_workspace.mon_*): p(y|x0) under the parameters BEFORE the
    step's update -- from the v0.W + d the positive-phase launch stores on the way (launch-per-phase path), or from a
    contraction of its own behind the one-launch chain (mode 2) -- and the minibatch's two loss terms added to the float64
    running sums: every bit against the oracle; the step itself is unchanged by the monitor."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 500 + V + H)
    v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 13)
    prev = ops.fused_chain_mode(mode)
    try:
        outs = []
        for with_mon in (True, False):
            P = P_of(ops, W, b, d, U, c)
            G = GradSet(V, H, C, ops.device)
            sums = torch.tensor([0.25, 4.0], dtype=torch.float64, device=ops.device)
            lab = dev(ops, labels.astype(np.int32))
            ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, 11, 5, 32, None, None, True, 0.05, 1.0, float(B), 0.0,
                        monitor=(lab, sums) if with_mon else None)
            ops.check_chain_status()
            outs.append([t.clone() for t in (P.W, P.b, P.d, P.U, P.c)])
            if with_mon:
                pre_o, _, p_o = oracle.rbm_free_energy(v_bin, W, d, U, c)
                rl, bl = oracle.rbm_class_loss(p_o, labels)
                mon = ops._cd_ws[1][7]
                assert_bitwise(mon[0], pre_o, "x.W + d stored by the positive-phase launch")
                assert_bitwise(mon[1], p_o, "p(y|x0) before the update")
                assert_bitwise(mon[2], rl, "per-row loss terms")
                assert sums.tolist() == [0.25 + float(bl[0]), 4.0 + float(bl[1])]
            else:
                assert sums.tolist() == [0.25, 4.0]
        ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, 11, 5, 32)
        Wn, bn, dn, Un, cn = (a.copy() for a in (W, b, d, U, c))
        oracle.rbm_update(Wn, bn, dn, Un, cn, *oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk), 0.05, 1.0, float(B), 0.0)
        for tag, out in zip(("with the monitor", "without"), outs):
            for got, want, nm in zip(out, (Wn, bn, dn, Un, cn), "WbdUc"):
                assert_bitwise(got, want, f"{nm} after the step {tag}")
    finally:
        ops.fused_chain_mode(prev)


@pytest.mark.parametrize("V,H,C,B,k,pcd", [(784, 512, 10, 256, 1, False), (784, 512, 10, 256, 3, True), (784, 128, 2, 64, 2, False),
                                           (1024, 512, 4, 32, 2, True), (64, 32, 12, 96, 1, False), (784, 512, 10, 224, 2, False)])
def test_fused_chain_kernel_equals_oracle_and_launch_per_phase(ops, oracle, V, H, C, B, k, pcd):
    """The one-launch Gibbs chain (k_chain: a row block of 32 per XCD, XCD-local flag barriers between the phases, states
    handed over through the XCD's L2 with sc1 loads) against the oracle AND against the launch-per-phase path (a workspace
    without the sync buffer), bit for bit, twice in a row on the same workspace (the barrier rounds carry over from launch
    to launch); its status word stays 0.  Shapes: the benchmark's, PCD, fewer than 8 row blocks, every workgroup holding a
    visible strip (V = 1024: the class conditional runs behind the strips), the largest class count, a ragged count of blocks."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 900 + V + H)
    v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 12)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    ws_f, ws_c = ops.cd_workspace(P, B), ops.cd_workspace(P, B, fused=False)
    assert ws_f[5] is not None and ws_c[5] is None
    cv0, cy0 = np.roll(v_bin, 1, axis=0).astype(np.uint8), np.roll(y1h, 1, axis=0).copy()
    chains = {n: ((dev(ops, cv0), dev(ops, cy0)) if pcd else (None, None)) for n in ("f", "c")}
    ocv, ocy = (cv0.astype(np.float32), cy0.copy()) if pcd else (None, None)
    v0, y0 = dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h)
    prev = ops.fused_chain_mode(2)      # the fused kernel for every shape it takes (default: from k = 2 on)
    try:
        _fused_chain_rounds(ops, oracle, P, G, ws_f, ws_c, chains, ocv, ocy, v_bin, y1h, v0, y0, W, b, d, U, c, k, pcd, B)
    finally:
        ops.fused_chain_mode(prev)
    assert int(ws_f[5][512].item()) == 0


def _fused_chain_rounds(ops, oracle, P, G, ws_f, ws_c, chains, ocv, ocy, v_bin, y1h, v0, y0, W, b, d, U, c, k, pcd, B):
    for rep in range(2):
        seed, off, row0 = 17, 300 + 50 * rep, 64
        ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, seed, off, row0, ocv, ocy)
        outs = {}
        for n, ws in (("f", ws_f), ("c", ws_c)):
            outs[n] = ops.cd_step(P, G, v0, y0, k, seed, off, row0, chains[n][0], chains[n][1], False, 0.05, 1.0, float(B), 0.0, ws)
        ops.check_chain_status(ws_f)
        for n in ("f", "c"):
            o = outs[n]
            assert_bitwise(o[0], ph0, f"p(h0) [{n}, rep {rep}]")
            assert_bitwise(o[4], phk, f"p(h_k) [{n}, rep {rep}]")
            if pcd:
                assert np.array_equal(chains[n][0].cpu().numpy(), ocv.astype(np.uint8)), f"persistent v [{n}, rep {rep}]"
                assert_bitwise(chains[n][1], ocy, f"persistent y [{n}, rep {rep}]")
            else:
                assert np.array_equal(o[2].cpu().numpy(), vk.astype(np.uint8)), f"v_k [{n}, rep {rep}]"
                assert_bitwise(o[3], yk, f"y_k [{n}, rep {rep}]")


def test_config3_pcd10_full_size_carried_state(ops, oracle):
    """BASELINE.json configs[2] exactly: 784v x 512h, C = 10, CD-10 with persistent chains, batch 256.  Three consecutive
    steps with the chains AND the parameters carried from step to step; every sampled bit, every probability and every
    parameter is compared with the oracle after every step."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    V, H, C, B, k = 784, 512, 10, 256, 10
    W, b, d, U, c = synth_params(V, H, C, 33)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    ws = ops.cd_workspace(P, B)
    v_init, _, _, lab_init, y_init = synth_inputs(B, V, H, C, 50)
    cv_o, cy_o = v_init.copy(), y_init.copy()                      # oracle-side chains (updated in place)
    cv, cy = dev(ops, v_init.astype(np.uint8)), dev(ops, y_init)
    seed, lr, sparse = 4242, 0.05, 0.0001
    for step in range(3):
        v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 60 + step)
        off, row0 = (3 * k + 1) * step, 256 * step
        out = ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, seed, off, row0, cv, cy, True, lr, 1.0,
                          float(B), sparse, ws)
        assert out[2] is cv and out[3] is cy                        # PCD: (v_k, y_k) are written straight into the chains
        ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, seed, off, row0, cv_o, cy_o)
        g = oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
        oracle.rbm_update(W, b, d, U, c, *g, lr, 1.0, float(B), sparse)
        assert_bitwise(ws[0], ph0, f"p(h0), step {step}")
        assert np.array_equal(cv.cpu().numpy(), vk.astype(np.uint8)), f"v_k, step {step}"
        assert_bitwise(cy, yk, f"y_k, step {step}")
        assert_bitwise(ws[4], phk, f"p(h_k), step {step}")
        assert np.array_equal(cv.cpu().numpy(), cv_o.astype(np.uint8)), f"persistent v, step {step}"
        assert_bitwise(cy, cy_o, f"persistent y, step {step}")
        for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
            assert_bitwise(got, want, f"{nm} after step {step}")
        for got, want, nm in ((G.gb, g[1], "gb"), (G.gd, g[2], "gd"), (G.gU, g[3], "gU"), (G.gc, g[4], "gc")):
            assert_bitwise(got, want, f"{nm}, step {step}")
    assert not np.array_equal(cv_o, v_init)                         # the chains moved


@pytest.mark.parametrize("V,H,C,B,k", [(784, 512, 10, 256, 1), (784, 128, 2, 128, 2), (13, 9, 3, 5, 1)])
def test_data_parallel_code_path_on_device(ops, oracle, V, H, C, B, k):
    """What one rank of the data-parallel step runs (BASELINE.json configs[3]): cd_step(fuse=0) materialises ALL five
    gradient blocks, (the all-reduce: identity on one rank), then qbm_rbm_update applies them -- bit-exact against the
    oracle's update.  Then the two-rank arithmetic: half-batches at row0 = 0 / B/2, gradients summed (what the all-reduce
    does), one update with the global batch size -- chains bit-identical to the whole batch, parameters within fp32
    rounding of the single-batch result (the batch sum is associated per rank)."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 700 + V + H)
    v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 12)
    seed, off, row0, lr, sparse = 7, 40, 2048, 0.05, 0.001
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    v8 = dev(ops, v_bin.astype(np.uint8))
    ws = ops.cd_step(P, G, v8, dev(ops, y1h), k, seed, off, row0, None, None, False, lr, 1.0, float(B), sparse)
    for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
        assert_bitwise(got, want, f"{nm} untouched by cd_step(fuse=0)")
    ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, seed, off, row0)
    g = oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
    for got, want, nm in zip((G.gW, G.gb, G.gd, G.gU, G.gc), g, ("gW", "gb", "gd", "gU", "gc")):
        assert_bitwise(got, want, nm)
    ops.update(P, G, lr, 1.0, float(B), sparse)                     # with_gb=True: all five blocks
    Wn, bn, dn, Un, cn = (a.copy() for a in (W, b, d, U, c))
    oracle.rbm_update(Wn, bn, dn, Un, cn, *g, lr, 1.0, float(B), sparse)
    for got, want, nm in ((P.W, Wn, "W"), (P.b, bn, "b"), (P.d, dn, "d"), (P.U, Un, "U"), (P.c, cn, "c")):
        assert_bitwise(got, want, f"{nm} after qbm_rbm_update")
    if B < 2:
        return
    # two "ranks"
    half = B // 2
    P2 = P_of(ops, W, b, d, U, c)
    Gs = [GradSet(V, H, C, ops.device) for _ in range(2)]
    vks = []
    for r, (lo, hi) in enumerate(((0, half), (half, B))):
        w2 = ops.cd_step(P2, Gs[r], v8[lo:hi].contiguous(), dev(ops, y1h[lo:hi]), k, seed, off, row0 + lo, None, None, False,
                         lr, 1.0, float(hi - lo), sparse)
        vks.append(w2[2].clone())
    assert torch.equal(torch.cat(vks), ws[2])                        # sharded chains == the whole batch's chains, bit for bit
    Gs[0].flat += Gs[1].flat                                          # the all-reduce
    ops.update(P2, Gs[0], lr, 1.0, float(B), sparse)
    for got, want, nm in ((P2.W, Wn, "W"), (P2.b, bn, "b"), (P2.d, dn, "d"), (P2.U, Un, "U"), (P2.c, cn, "c")):
        err = np.max(np.abs(got.cpu().numpy() - want))
        assert err <= 2e-6, (nm, err)


def test_empty_shard_zeroes_gradients(ops):
    """B == 0 (a data-parallel rank whose shard of a short last minibatch is empty): the gradient blocks are zeroed, so the
    all-reduce adds nothing from this rank, and the parameters stay as they are."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    V, H, C = 784, 128, 2
    W, b, d, U, c = synth_params(V, H, C, 3)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    e = lambda n, dt=torch.float32: torch.empty(0, n, dtype=dt, device=ops.device)
    for call in ("cd", "disc"):
        G.flat.fill_(3.0)
        if call == "cd":
            ops.cd_step(P, G, e(V, torch.uint8), e(C), 1, 1, 0, 0, None, None, False, 0.05, 1.0, 1.0, 0.0)
            assert not G.flat[:-1].any()
        else:
            p = ops.disc_step(P, G, e(V), torch.empty(0, dtype=torch.int32, device=ops.device), False, False, 0.05, 1.0,
                              1.0, 0.0)
            assert p.shape == (0, C)
            assert not G.gW.any() and not G.gd.any() and not G.gU.any() and not G.gc.any()
            assert (G.gb == 3.0).all()                                # the discriminative step never owns gb (:138)
        assert float(G.count) == 3.0                                  # the row-count slot belongs to the caller
    assert_bitwise(P.W, W, "W untouched")


def test_sharding_invariance_full_size(ops):
    """Size-independent property at the benchmark shape: sampling rows [0,256) in one call equals two
    calls on the halves with row0 = 0 / 128 (what two data-parallel ranks would do), bit for bit."""
    V, H, C, B = 784, 512, 10, 256
    W, b, d, U, c = synth_params(V, H, C, 5)
    v_bin, _, h_bin, labels, y1h = synth_inputs(B, V, H, C, 11)
    P = P_of(ops, W, b, d, U, c)
    v, y, h = dev(ops, v_bin), dev(ops, y1h), dev(ops, h_bin)
    for fn, args in ((ops.sample_hidden, (v, y)), (ops.sample_visible, (h,)), (ops.sample_class, (h,))):
        p, s = fn(P, *args, 17, 3, 0, True, True)
        halves = [fn(P, *(a[i:i + 128] for a in args), 17, 3, i, True, True) for i in (0, 128)]
        assert torch.equal(p, torch.cat([t[0] for t in halves]))
        assert torch.equal(s, torch.cat([t[1] for t in halves]))
    p2, s2 = ops.sample_hidden(P, v, y, 17, 3, 0, True, True)
    p1, s1 = ops.sample_hidden(P, v, y, 17, 3, 0, True, True)
    assert torch.equal(p1, p2) and torch.equal(s1, s2)           # idempotent / deterministic
    _, s3 = ops.sample_hidden(P, v, y, 17, 4, 0, True, True)
    assert not torch.equal(s1, s3)                               # a new draw id gives new bits
    assert abs(float(s1.mean()) - float(p1.mean())) < 5e-3       # Bernoulli mean matches the probabilities


def test_edge_cases(ops):
    from qbm_image_classification_amd._lib import QbmHipError
    from qbm_image_classification_amd.rbm_ops import ParamSet

    W, b, d, U, c = synth_params(16, 8, 2, 1)
    P = P_of(ops, W, b, d, U, c)
    p, s = ops.sample_hidden(P, torch.zeros(0, 16, device=ops.device), None, 0, 0, 0, True, True)  # empty batch
    assert p.shape == (0, 8) and s.shape == (0, 8)
    import ctypes as C

    ps = P.struct()
    rc = ops.lib.qbm_rbm_sample_hidden(C.byref(ps), None, None, 4, None, None, 0, 0, 0, None)  # null buffers
    assert rc == -1 and b"null" in ops.lib.qbm_last_error()
    rc = ops.lib.qbm_rbm_sample_hidden(C.byref(ps), None, None, -3, None, None, 0, 0, 0, None)
    assert rc == -1
    big = synth_params(8, 4, 40, 2)  # C > QBM_MAX_CLASSES
    with pytest.raises(QbmHipError):
        ops.sample_class(P_of(ops, *big), torch.zeros(2, 4, device=ops.device))
    # saturated activations: probabilities exactly 0/1, free energy overflows to inf like the reference
    Wb = np.full((16, 8), 50.0, np.float32)
    Pb = P_of(ops, Wb, b, d, U, c)
    ph, sh = ops.sample_hidden(Pb, torch.ones(3, 16, device=ops.device), None, 1, 1, 0, True, True)
    assert (ph == 1).all() and (sh == 1).all()
    nf, _ = ops.free_energy(Pb, torch.ones(3, 16, device=ops.device))
    assert torch.isinf(nf).all()


def test_qubo_gibbs_bitwise(oracle):
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(3)
    # fewer than 64 chains or n > 276: J rows from L2 (k_qubo_gibbs); otherwise the LDS-resident triangle
    # (k_qubo_gibbs_lds, persistent workgroups: more chains than wavefronts makes every wave walk several)
    for (B, R, n, sweeps, b0, b1) in [(3, 5, 7, 4, 1.0, 1.0), (2, 10, 65, 6, 0.1, 3.0), (4, 7, 257, 3, 1.0, 1.0),
                                      (1, 3, 300, 2, 0.5, 2.0), (2, 4, 1, 3, 1.0, 1.0),
                                      (8, 16, 257, 3, 0.2, 2.0), (4, 32, 65, 5, 1.0, 1.0), (2, 40, 276, 2, 0.5, 1.5),
                                      (70, 1, 7, 4, 1.0, 1.0), (150, 40, 33, 2, 0.3, 3.0), (2, 40, 277, 2, 0.5, 1.5),
                                      (1, 70, 700, 2, 0.7, 1.3),
                                      # more chains than wavefronts: every wavefront walks several chains, one after the other
                                      (45, 100, 130, 3, 0.3, 2.0), (90, 100, 65, 2, 0.5, 1.5), (50, 100, 257, 2, 0.2, 2.5),
                                      # n = 64 k + 1: k whole tiles + the last spin as a wave-uniform scalar (round 4), both readers
                                      (3, 30, 129, 4, 0.3, 2.0), (2, 10, 193, 3, 1.0, 1.0), (3, 40, 193, 3, 0.2, 3.0), (1, 9, 321, 2, 0.5, 1.5),
                                      (2, 5, 385, 2, 0.5, 1.5)]:
        lin = rs.standard_normal((B, n)).astype(np.float32)
        Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * 0.3, 1)
        J = (Ju + Ju.T).astype(np.float32)
        st, mean = q.gibbs(torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda(), R, sweeps, b0, b1, 77, 5)
        so, mo = oracle.qubo_gibbs(lin, J, R, sweeps, b0, b1, 77, 5)
        assert np.array_equal(st.cpu().numpy(), so), f"states differ for n={n}"
        assert np.array_equal(mean.cpu().numpy(), mo), "mean over reads (float64: exact count / R)"


def _sparse_qubo(rs, n, deg, scale=0.5, hub=None):
    """symmetric J [n,n] with about `deg` neighbours per spin (+ one spin coupled to `hub` others)"""
    J = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in rs.choice(n, size=max(1, deg // 2), replace=False):
            if i != j:
                J[i, j] = J[j, i] = np.float32(rs.standard_normal() * scale)
    if hub:
        h = n // 3
        for j in rs.choice(n, size=hub, replace=False):
            if j != h:
                J[h, j] = J[j, h] = np.float32(rs.standard_normal() * scale)
    return J


def test_qubo_gibbs_csr_bitwise(oracle):
    """The sparse-coupling sampler (CSR rows and the chains' fields in LDS, ds_add_f32 into the neighbours' fields: round 4)
    against the oracle and against the dense kernel, bit for bit: one to eight wavefronts per workgroup (the chain count
    picks), rows of more than 64 neighbours, a ragged last tile, every tile count the kernel is built for (n = 60 ... 1024),
    an empty row, chain0 (sharded minibatches); couplings and fields in the DENORMAL range (the LDS atomic must neither flush
    nor round them differently from fmaf(+-1, J, F)); the overflow protocol (more couplings than `cap`: status 1, nothing
    sampled, the dense kernel takes over)."""
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(12)
    for (B, R, n, deg, hub, sweeps, b0, b1, chain0) in [(1, 5, 300, 8, None, 4, 0.5, 2.0, 0), (2, 40, 384, 6, 100, 3, 0.3, 3.0, 0),
                                                        (1, 100, 717, 8, 40, 3, 0.2, 2.5, 0), (3, 200, 520, 10, 70, 2, 0.5, 1.5, 0),
                                                        (1, 9, 1024, 12, 200, 2, 1.0, 1.0, 0), (2, 6, 60, 4, 30, 5, 0.4, 2.0, 0),
                                                        (30, 100, 330, 6, None, 2, 0.3, 2.0, 0), (2, 33, 650, 8, 65, 3, 0.3, 2.0, 17),
                                                        (1, 12, 897, 8, None, 2, 0.5, 1.5, 0),
                                                        # no sweep at all / one sweep (the helper wavefront's first and only buffer),
                                                        # 256 chains (one per workgroup, with the helper) and 257 (two per workgroup)
                                                        (1, 7, 300, 8, None, 0, 0.5, 1.5, 0), (1, 7, 300, 8, None, 1, 0.5, 1.5, 0),
                                                        (256, 1, 290, 6, None, 3, 0.4, 2.0, 0), (257, 1, 290, 6, None, 3, 0.4, 2.0, 0)]:
        lin = rs.standard_normal((B, n)).astype(np.float32)
        J = _sparse_qubo(rs, n, deg, hub=hub)
        J[5, :] = 0
        J[:, 5] = 0                                      # a spin without neighbours
        lt, Jt = torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda()
        st, mean, head = q.gibbs_csr(lt, Jt, R, sweeps, b0, b1, 77, 5, True, chain0)
        assert head.cpu().tolist() == [0, int(np.count_nonzero(J))]
        sd, md = q.gibbs(lt, Jt, R, sweeps, b0, b1, 77, 5, True, chain0)
        assert torch.equal(st, sd) and torch.equal(mean, md), f"sparse and dense kernels differ for n={n}"
        if B * R * n <= 200000:
            so, mo = oracle.qubo_gibbs(lin, J, R, sweeps, b0, b1, 77, 5, chain0)
            assert np.array_equal(st.cpu().numpy(), so), f"states differ for n={n}"
            assert np.array_equal(mean.cpu().numpy(), mo)
    # denormal couplings and fields (beta large enough for them to decide spins)
    n = 320
    J = (_sparse_qubo(rs, n, 8, scale=1.0).astype(np.float64) * 2.5e-39).astype(np.float32)
    lin = (rs.standard_normal((2, n)) * 2.5e-39).astype(np.float32)
    assert J[J != 0].size > 2000 and np.abs(J[J != 0]).max() < 1.17549435e-38
    st, _, head = q.gibbs_csr(torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda(), 6, 5, 3e38, 3e38, 4, 2)
    so, _ = oracle.qubo_gibbs(lin, J, 6, 5, 3e38, 3e38, 4, 2)
    sz, _ = oracle.qubo_gibbs(lin * 0, J * 0, 6, 5, 3e38, 3e38, 4, 2)
    assert not np.array_equal(so, sz), "(the denormal terms decide spins: flushing them would show)"
    assert head[0].item() == 0 and np.array_equal(st.cpu().numpy(), so), "denormal couplings"
    # overflow: a dense matrix does not fit; nothing is written, the dense kernel is the fallback
    n = 400
    Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * 0.1, 1)
    J = torch.from_numpy((Ju + Ju.T).astype(np.float32)).cuda()
    lin = torch.from_numpy(rs.standard_normal((1, n)).astype(np.float32)).cuda()
    lim = q.lib.qbm_qubo_csr_max_entries(n, 8)
    assert 15000 < lim < n * (n - 1)
    st, mean, head = q.gibbs_csr(lin, J, 8, 2, 1.0, 1.0, 1, 0)
    assert head.cpu().tolist() == [1, n * (n - 1)]
    # (through the C ABI with buffers of its own: neither the states nor the means are written by a launch that gave up)
    from qbm_image_classification_amd._lib import check, current_stream, ptr
    ws = torch.zeros(q.lib.qbm_qubo_csr_workspace_bytes(n, lim) // 4, dtype=torch.int32, device="cuda")
    st_keep = torch.full((1, 8, n), 7, dtype=torch.uint8, device="cuda")
    mean_keep = torch.full((1, n), -3.0, dtype=torch.float64, device="cuda")
    check(q.lib.qbm_qubo_csr_build(ptr(J), n, lim, ptr(ws), current_stream()), "qbm_qubo_csr_build")
    check(q.lib.qbm_qubo_gibbs_csr(ptr(lin), ptr(ws), lim, 1, 8, n, 2, None, 1.0, 1.0, 1, 0, 0, ptr(st_keep), ptr(mean_keep), current_stream()),
          "qbm_qubo_gibbs_csr")
    assert ws[0].item() == 1 and (st_keep == 7).all() and (mean_keep == -3.0).all()
    st2, _, head2 = q.gibbs_csr(lin, J * (torch.rand_like(J) < 0.0), 8, 2, 1.0, 1.0, 1, 0, cap=0)   # no couplings at all, cap 0
    assert head2.cpu().tolist() == [0, 0]
    so, _ = oracle.qubo_gibbs(lin.cpu().numpy(), np.zeros((n, n), np.float32), 8, 2, 1.0, 1.0, 1, 0)
    assert np.array_equal(st2.cpu().numpy(), so)
    with pytest.raises(Exception):
        q.gibbs_csr(lin, J, 8, 2, 1.0, 1.0, 1, 0, cap=lim + 1)


def test_qubo_gibbs_sharding_invariance(oracle):
    """Rows [lo, hi) of a minibatch sampled with chain0 = lo * R are exactly the rows [lo, hi) of the whole call (what a
    data-parallel rank of Disc_QBM does), on both kernels (LDS-resident triangle and J rows from L2)."""
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(8)
    for (B, R, n, sweeps) in [(73, 20, 257, 5), (9, 3, 40, 6), (6, 40, 300, 3)]:
        lin = rs.standard_normal((B, n)).astype(np.float32)
        Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * 0.2, 1)
        J = torch.from_numpy((Ju + Ju.T).astype(np.float32)).cuda()
        whole, mean = q.gibbs(torch.from_numpy(lin).cuda(), J, R, sweeps, 0.3, 2.0, 5, 9)
        lo, hi = B // 3, B - 2
        part, pmean = q.gibbs(torch.from_numpy(lin[lo:hi]).cuda(), J, R, sweeps, 0.3, 2.0, 5, 9, True, lo * R)
        assert torch.equal(part, whole[lo:hi]) and torch.equal(pmean, mean[lo:hi])
        so, _ = oracle.qubo_gibbs(lin[lo:hi], J.cpu().numpy(), R, sweeps, 0.3, 2.0, 5, 9, lo * R)
        assert np.array_equal(part.cpu().numpy(), so)


def test_qbm_linear_fields_bitwise(oracle):
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    g = golden("qbm_small.npz")
    X, Y = g["X"], g["Y"]
    for labels in (None, Y):
        lin = q.linear_fields(X, labels, g["W_vh"], g["W_vo"], g["b_h"], g["b_o"], float(g["beta_eff"]))
        want = oracle.qbm_linear_fields(X, labels, g["W_vh"], g["W_vo"], g["b_h"], g["b_o"], float(g["beta_eff"]))
        assert_bitwise(lin, want, "linear fields")
        Q = g["Q_clamped"] if labels is not None else g["Q_unclamped"]
        diag = np.stack([np.diag(Q[i]) for i in range(len(X))])
        assert np.max(np.abs(lin.cpu().numpy() - diag)) <= 1e-6 * max(1.0, np.abs(diag).max())  # vs the reference's Q
    # shapes around the kernel's blocking (64 variables x 4 images per workgroup, chunks of 56 terms through LDS): ragged in
    # every direction, one chunk, many chunks, the benchmark's; gray-scale pixels and negative weights (no exact products)
    rs = np.random.RandomState(4)
    for (B, D, H) in [(1, 1, 1), (3, 55, 63), (4, 56, 64), (5, 57, 65), (9, 113, 130), (73, 784, 256), (2, 300, 257), (6, 8, 700)]:
        X = rs.random_sample((B, D))
        Y = rs.randint(0, 2, B)
        W_vh, W_vo = rs.standard_normal((1 + D, H)) * 0.3, rs.standard_normal(D) * 0.3
        b_h, b_o = rs.standard_normal(H), np.array([rs.standard_normal()])
        for labels in (None, Y):
            lin = q.linear_fields(X, labels, W_vh, W_vo, b_h, b_o, 0.7)
            assert_bitwise(lin, oracle.qbm_linear_fields(X, labels, W_vh, W_vo, b_h, b_o, 0.7), f"linear fields {B}x{D}x{H}")


def test_qbm_batch_stats_bitwise(oracle):
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    g = golden("qbm_small.npz")
    X, Y, H = g["X"], g["Y"], int(g["H"])
    for tag, S, labels in (("clamped", g["S_clamped"], Y), ("unclamped", g["S_unclamped"], None)):
        mean = S.astype(np.float64).mean(axis=1)
        got = q.batch_stats(torch.from_numpy(X), labels, torch.from_numpy(mean), H)
        want = oracle.qbm_batch_stats(X, labels, mean, H)
        for a, b, nm in zip(got, want, ("W_vh", "W_vo", "b_h", "b_o")):
            assert np.array_equal(a.cpu().numpy(), b), f"{tag} {nm}"                       # float64, bit for bit
        assert np.max(np.abs(got[0].cpu().numpy() - g[f"avg_{tag}_W_vh"])) <= 1e-12          # and vs the reference
    rs = np.random.RandomState(1)
    Xb = (rs.random_sample((73, 784)) < 0.13).astype(np.float64)
    Yb = rs.randint(0, 2, 73)
    mb = rs.randint(0, 101, size=(73, 256)).astype(np.float64) / 100.0
    got = q.batch_stats(torch.from_numpy(Xb), Yb, torch.from_numpy(mb), 256)
    want = oracle.qbm_batch_stats(Xb, Yb, mb, 256)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)


def test_gibbs_vs_exact_enumeration_on_gpu():
    import itertools

    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(12)
    n, R = 8, 40000
    lin = rs.standard_normal((2, n)).astype(np.float32)
    Ju = np.triu(rs.standard_normal((n, n)).astype(np.float32) * 0.7, 1)
    J = (Ju + Ju.T).astype(np.float32)
    _, mean = q.gibbs(torch.from_numpy(lin), torch.from_numpy(J), R, 30, 1.0, 1.0, 5, 0)
    states = np.array(list(itertools.product([0, 1], repeat=n)), dtype=np.float64)
    for b in range(2):
        E = states @ lin[b].astype(np.float64) + 0.5 * np.einsum("si,ij,sj->s", states, J.astype(np.float64), states)
        w = np.exp(-(E - E.min())); w /= w.sum()
        assert np.max(np.abs(mean[b].cpu().numpy() - w @ states)) < 0.012


def test_qubo_full_size_properties():
    """Config #5 shape (73 images x 100 reads, n = 257): size-independent properties instead of the oracle."""
    from qbm_image_classification_amd.sampler import HipQuboOps, auto_beta_range

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(2)
    B, R, n = 73, 100, 257
    lin = (rs.standard_normal((B, n)) * 3).astype(np.float32)
    Ju = np.triu(rs.uniform(-1, 1, (n, n)).astype(np.float32), 1)
    J = (Ju + Ju.T).astype(np.float32)
    lt, Jt = torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda()
    b0, b1 = auto_beta_range(lin, J)
    s1, m1 = q.gibbs(lt, Jt, R, 50, b0, b1, 3, 0)
    s2, m2 = q.gibbs(lt, Jt, R, 50, b0, b1, 3, 0)
    assert torch.equal(s1, s2) and torch.equal(m1, m2)                          # deterministic
    assert np.array_equal(m1.cpu().numpy(), s1.cpu().numpy().sum(axis=1, dtype=np.int64) / float(R))   # mean == count / R, correctly rounded
    sf = s1.float()
    E = lambda s: (s * lt[:, None, :]).sum(-1) + 0.5 * torch.einsum("bri,ij,brj->br", s, Jt, s)
    s0, _ = q.gibbs(lt, Jt, R, 0, b0, b1, 3, 0)                                  # 0 sweeps = the random start state
    assert E(sf).mean() < E(s0.float()).mean() - 50.0                            # annealing lowers the energy
    half = q.gibbs(lt[:36], Jt, R, 50, b0, b1, 3, 0)[0]
    assert torch.equal(half, s1[:36])                                            # images are independent chains


def test_disc_qbm_trains_on_gpu(tmp_path):
    """End-to-end Disc_QBM(solver="HIP"): a linearly separable toy problem is learnt."""
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    rs = np.random.RandomState(0)
    D, N = 16, 96
    Y = rs.randint(0, 2, N)
    X = (rs.random_sample((N, D)) < 0.1).astype(np.float64)
    X[Y == 1, :8] = (rs.random_sample(((Y == 1).sum(), 8)) < 0.8)
    X[Y == 0, 8:] = (rs.random_sample(((Y == 0).sum(), 8)) < 0.8)
    np.random.seed(1)
    q = Disc_QBM(dim_input=D, num_classes=2, epochs=6, n_hidden_nodes=6, seed=3, solver="HIP", sample_count=50,
                 anneal_steps=30, beta_eff=1.0, speicherort=str(tmp_path) + "/", param_string="toy")
    q.train_model(X, Y, X, Y, batch_size=16, learning_rate=0.3)
    acc = (q.predict_batch(X) == Y).mean()
    assert acc > 0.9, acc
    assert len(q.training_history.acc_per_epoch) == 6 and (tmp_path / "toy" / "e6_toy.pkl").exists()
    p, outs = q.predict(X[0])
    assert p in (0, 1) and outs.shape == (50,)
    s = q.get_samples(X[0], label=1)
    assert len(s) == 50 and set(s[0].keys()) == set(range(6))


def test_qbm_device_state_kernels_bitwise(oracle):
    """The kernels behind the device-resident Disc_QBM step (couplings + coefficient statistics, per-image annealing range,
    statistics difference, update) against the oracle: float64 / fp32 values bit for bit."""
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(21)
    # (the statistics kernel works in tiles of 4 rows x 256 columns and chunks of 128 images: shapes on both sides of each)
    for D, H, B, restricted, beta_eff in ((12, 5, 4, False, 0.7), (784, 256, 73, False, 1.0), (784, 33, 9, True, 2.0), (7, 300, 131, False, 1.0),
                                          (3, 256, 128, False, 0.5), (9, 257, 1, True, 1.0)):
        n_theta = (1 + D) * H + D + H + 1
        theta = rs.uniform(-1, 1, n_theta)
        W_hh = None if restricted else np.triu(rs.uniform(-1, 1, (H, H)), 1)
        X = (rs.random_sample((B, D)) < 0.2).astype(np.float64)
        Y = rs.randint(0, 2, B).astype(np.int32)
        t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
        th = t64(theta)
        whh = None if W_hh is None else t64(W_hh)
        W_vh = theta[:(1 + D) * H].reshape(1 + D, H)
        for clamped in (True, False):
            J, rowabs, rowmin = q.couplings(th, whh, H, clamped, beta_eff)
            Jo = oracle.qbm_couplings(W_vh, W_hh, clamped, beta_eff)
            ra, rm = oracle.qbm_coupling_stats(Jo)
            assert np.array_equal(J.cpu().numpy(), Jo)
            assert np.array_equal(rowabs.cpu().numpy(), ra) and np.array_equal(rowmin.cpu().numpy(), rm)
            labels = torch.from_numpy(Y).cuda() if clamped else None
            lin = q.linear_fields_dev(t64(X), labels, th, D, H, beta_eff)
            W_vo, b_h, b_o = theta[(1 + D) * H:(1 + D) * H + D], theta[(1 + D) * H + D:-1], theta[-1:]
            lo = oracle.qbm_linear_fields(X, Y if clamped else None, W_vh, W_vo, b_h, b_o, beta_eff)
            assert np.array_equal(lin.cpu().numpy().view(np.uint32), lo.view(np.uint32))
            beta = q.beta_range(lin, rowabs, rowmin)
            bo = oracle.qbm_beta_range(lo, ra, rm)
            assert np.array_equal(beta.cpu().numpy().view(np.uint32), bo.view(np.uint32)), (D, H, clamped)
            if not restricted or not clamped:
                assert (bo[:, 1] >= 10 * bo[:, 0] * (1 - 1e-6)).all() and (bo[:, 0] > 0).all()
            R, sweeps = 6, 4
            if D * H < 5000 or clamped:
                st, mean = q.gibbs_dev(lin, J, R, sweeps, beta, 11, 3, 40)
                for b in range(B):
                    so, mo = oracle.qubo_gibbs(lo[b:b + 1], Jo, R, sweeps, float(bo[b, 0]), float(bo[b, 1]), 11, 3, 40 + b * R)
                    assert np.array_equal(st[b:b + 1].cpu().numpy(), so) and np.array_equal(mean[b:b + 1].cpu().numpy(), mo)
        mc = rs.randint(0, 11, (B, H)) / 10.0
        mu = rs.randint(0, 11, (B, H + 1)) / 10.0
        out = torch.zeros(n_theta + 1, dtype=torch.float64, device="cuda")
        q.stats_diff(t64(X), torch.from_numpy(Y).cuda(), t64(mc), t64(mu), D, H, out)
        want = oracle.qbm_stats_diff(X, Y, mc, mu, H)
        assert np.array_equal(out[:-1].cpu().numpy(), want) and float(out[-1]) == 0.0
        # ... and with the NLL sum from the same launch (compute_nll's term, faster_dqbm.py:986-991: float32 log, a monitor)
        out2 = torch.zeros(n_theta + 1, dtype=torch.float64, device="cuda")
        q.stats_diff(t64(X), torch.from_numpy(Y).cuda(), t64(mc), t64(mu), D, H, out2, nll=True)
        probs = np.stack([1.0 - mu[:, 0], mu[:, 0]], axis=1).astype(np.float32)
        nll = float(np.sum(-np.log(probs + np.float32(1e-12))[np.arange(B), Y].astype(np.float64)))
        assert np.array_equal(out2[:-1].cpu().numpy(), want) and abs(float(out2[-1]) - nll) <= 1e-5 * max(1.0, abs(nll))
        q.apply_update(th, out, D, H, 0.05, float(B))
        assert np.array_equal(th.cpu().numpy(), theta - 0.05 * (want / float(B)))
    # a QUBO without any coefficient: the fallback range
    z = torch.zeros(2, 7, dtype=torch.float32, device="cuda")
    zz = torch.zeros(7, dtype=torch.float64, device="cuda")
    assert np.array_equal(q.beta_range(z, zz, zz).cpu().numpy(), np.array([[0.1, 1.0]] * 2, np.float32))


def test_disc_qbm_device_step_equals_oracle_double(oracle):
    """Disc_QBM.train_for_one_iteration with the parameters resident on the GPU (fields, couplings, annealing range, both
    sampling phases, statistics, update: all kernels) against the same class driven by the oracle-backed double: every
    parameter bit for bit after each of 3 iterations, on a 784-input model."""
    from oracle_backends import OracleQuboOps
    from qbm_image_classification_amd.disc_qbm import Disc_QBM
    from qbm_image_classification_amd.sampler import HipQuboOps

    rs = np.random.RandomState(5)
    X = (rs.random_sample((10, 784)) < 0.15).astype(np.float64)
    Y = rs.randint(0, 2, 10)

    def make(ops):
        np.random.seed(17)
        return Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=20, seed=9, solver="HIP", sample_count=8, anneal_steps=6,
                        beta_eff=1.0, ops=ops, data_parallel=False)

    g, o = make(HipQuboOps("cuda:0")), make(OracleQuboOps())
    for it in range(3):
        eg, ng = g.train_for_one_iteration(X, Y, 0.05)
        eo, no = o.train_for_one_iteration(X, Y, 0.05)
        # (the NLL is a sum of float32 logarithms, of 1e-12 when an output mean is 0 or 1: one ulp of such a term is 1.9e-6)
        assert np.array_equal(eg, eo) and abs(ng - no) <= 1e-5, (it, ng, no)
        for name in ("weights_all_visible_to_hidden", "weights_clamped_visible_to_output", "biases_hidden", "biases_output"):
            assert np.array_equal(getattr(g, name), getattr(o, name)), (it, name)
    assert np.array_equal(g.weights_hidden_hidden, o.weights_hidden_hidden)       # Q8: never learns
    assert np.array_equal(g.predict_batch(X), o.predict_batch(X))
    # assignment uploads: the next step starts from the assigned parameters on both sides
    g.biases_hidden = np.zeros(20); o.biases_hidden = np.zeros(20)
    g.train_for_one_iteration(X, Y, 0.05); o.train_for_one_iteration(X, Y, 0.05)
    assert np.array_equal(g.biases_hidden, o.biases_hidden) and np.array_equal(g.weights_all_visible_to_hidden, o.weights_all_visible_to_hidden)


def _cdqbm_analyse(g, Q, S):
    """energies s^T Q s, at-most-one rate and OR-link rate (pool unit == OR of its window) of reads S [R, n]"""
    W, pool0 = g["pool_windows"], int(g["pool"][0])
    Q, S = Q.astype(np.float64), S.astype(np.float64)
    E = np.einsum("ri,ij,rj->r", S, Q, S)
    amo = np.mean([[S[r, w].sum() <= 1 for w in W] for r in range(len(S))])
    link = np.mean([[S[r, pool0 + k] == (S[r, w].sum() > 0) for k, w in enumerate(W)] for r in range(len(S))])
    return E, amo, link


def test_qubo_from_Q_bitwise():
    """qbm_qubo_from_Q (what sample_Q does to its host matrix on the device: include/qbm_hip.h) against numpy, bit for bit:
    lin = float32(diag), J = float32(off + off.T) in Q's own precision, row sums of |J| (ascending, float64), the smallest
    non-zero magnitude per row, the "any quadratic term" flag; float64 and float32 matrices, sizes around the 64 x 64 tiles the
    kernel works in, entries below the diagonal too (the protocol says upper-triangular; the kernel does not rely on it)."""
    from qbm_image_classification_amd._lib import check, current_stream, ptr
    from qbm_image_classification_amd.sampler import HipQuboOps

    q = HipQuboOps("cuda:0")
    rs = np.random.RandomState(21)
    for n in (1, 2, 63, 64, 65, 130, 257, 717):
        for dt in (np.float64, np.float32):
            Q = (rs.standard_normal((n, n)) * (rs.random_sample((n, n)) < 0.3)).astype(dt)
            if n == 130:
                Q = np.triu(Q)
            Qd = torch.from_numpy(Q).cuda()
            f = lambda *sh, d=torch.float32: torch.empty(*sh, dtype=d, device="cuda")
            lin, J, ra, rm, flag = f(n), f(n, n), f(n, d=torch.float64), f(n, d=torch.float64), torch.zeros(1, dtype=torch.int32, device="cuda")
            check(q.lib.qbm_qubo_from_Q(ptr(Qd), int(dt == np.float64), n, ptr(lin), ptr(J), ptr(ra), ptr(rm), ptr(flag), current_stream()),
                  "qbm_qubo_from_Q")
            off = Q - np.diag(np.diag(Q))
            Jw = (off + off.T).astype(np.float32)
            assert np.array_equal(lin.cpu().numpy(), np.diag(Q).astype(np.float32)) and np.array_equal(J.cpu().numpy(), Jw), (n, dt)
            mag = np.abs(Jw).astype(np.float64)
            raw = np.zeros(n)
            for b in range(n):
                raw = raw + mag[:, b]                                  # ascending b, one add per term
            rmw = np.where(mag > 0, mag, np.inf).min(axis=1)
            rmw[~np.isfinite(rmw)] = 0.0
            assert np.array_equal(ra.cpu().numpy(), raw) and np.array_equal(rm.cpu().numpy(), rmw), (n, dt)
            assert int(flag.item()) == int(off.any()), (n, dt)
    Qz = torch.from_numpy(np.diag(rs.standard_normal(70))).cuda()
    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    f = lambda *sh, d=torch.float32: torch.empty(*sh, dtype=d, device="cuda")
    check(q.lib.qbm_qubo_from_Q(ptr(Qz), 1, 70, ptr(f(70)), ptr(f(70, 70)), ptr(f(70, d=torch.float64)), ptr(f(70, d=torch.float64)), ptr(flag),
                                current_stream()), "qbm_qubo_from_Q")
    assert int(flag.item()) == 0


def test_sample_q_on_reference_conv_qbm_qubo(oracle):
    """the sample_Q adapter on REAL Conv_Deep_QBM QUBOs -- the penalty-augmented (at-most-one per
    pooling window, OR-link to the pooling unit: src/qubo/builder.py:116-146) clamped / unclamped matrices that
    src/train/pipeline.py:20,35 hands to model.sampler.sample_Q, built by the imported reference for two MNIST digits
    (tools/make_golden_cdqbm.py; n = 716 / 717: 676 conv + 36 pooling + 4 dense units (+ 1 output)).
      * the kernel that serves n > 512 (12 tiles of 64 spins) equals the oracle bit for bit on these matrices;
      * annealed reads (1000 sweeps, the reference's default) reach the energy of a 5x longer CPU-oracle anneal within 0.5 %,
        and satisfy the OR-link constraint in >= 99 % of (read, window) pairs -- like the oracle's reads;
      * at-most-one: with the reference's penalty weight (0.8225) below the conv biases of background windows (-0.995 each)
        the QUBO's own optimum holds three units per such window, so the long oracle anneal violates it as well; the GPU
        reads' satisfaction rate is required to match the oracle's to 0.02 rather than to be high."""
    from qbm_image_classification_amd.sampler import HipGibbsSampler, HipQuboOps, auto_beta_range

    g = golden("cdqbm_qubo.npz")
    q = HipQuboOps("cuda:0")
    for name in ("Q_unclamped", "Q_clamped"):
        for img in range(2):
            Q = g[name][img]
            n = Q.shape[0]
            assert n in (716, 717) and not np.tril(Q, -1).any()
            lin = np.diag(Q).astype(np.float32)[None, :]
            off = Q - np.diag(np.diag(Q))
            J = (off + off.T).astype(np.float32)
            b0, b1 = auto_beta_range(lin, J)
            if img == 0:   # bit parity of the wide kernel on the real matrix
                st, _ = q.gibbs(torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda(), 8, 6, b0, b1, 3, 1)
                so, _ = oracle.qubo_gibbs(lin, J, 8, 6, b0, b1, 3, 1)
                assert np.array_equal(st.cpu().numpy(), so)
                # ... and so does the sparse kernel sample_Q picks for it (round 4: its 5762 couplings as CSR rows in LDS)
                sc, _, head = q.gibbs_csr(torch.from_numpy(lin).cuda(), torch.from_numpy(J).cuda(), 8, 6, b0, b1, 3, 1)
                assert head.cpu().tolist() == [0, int(np.count_nonzero(J))] and np.array_equal(sc.cpu().numpy(), so)
            smp = HipGibbsSampler(num_sweeps=1000, seed=7)
            S = smp.sample_Q(Q.astype(np.float64), 64)
            assert smp.last_kernel == "csr"
            if img == 0:   # the same samples from the dense kernel
                smd = HipGibbsSampler(num_sweeps=1000, seed=7)
                smd.sparse = False
                assert np.array_equal(smd.sample_Q(Q.astype(np.float64), 64), S) and smd.last_kernel == "dense"
            assert S.shape == (64, n) and S.dtype == np.float32 and set(np.unique(S)) <= {0.0, 1.0}
            E, amo, link = _cdqbm_analyse(g, Q, S)
            So, _ = oracle.qubo_gibbs(lin, J, 8, 5000, b0, b1, 5, 0)
            Eo, amo_o, link_o = _cdqbm_analyse(g, Q, So[0])
            assert E.mean() <= Eo.min() + 0.005 * abs(Eo.min()), (name, img, E.mean(), Eo.min())
            assert link >= 0.99 and link_o >= 0.99, (link, link_o)
            assert abs(amo - amo_o) <= 0.02, (amo, amo_o)


@pytest.mark.parametrize("V,H,C,B,k,pcd", [(784, 512, 10, 256, 1, False), (784, 512, 10, 256, 1, True), (784, 128, 2, 128, 1, False),
                                           (784, 128, 2, 512, 1, False), (64, 128, 12, 96, 1, False), (784, 512, 10, 48, 1, True),
                                           (784, 256, 4, 320, 1, False)])
def test_fused_tail_equals_oracle_and_launch_per_phase(ops, oracle, V, H, C, B, k, pcd):
    """The fused tail of a CD step (k_hgrad: last hidden phase, XCD-local barrier among the workgroups of a column group,
    two-phase gradient over 128-row W strips, update -- ONE launch) against the oracle and against the launch-per-phase
    path, bit for bit: three steps in a row on the same workspace with the parameters (and the persistent chains) carried
    over (the barrier rounds carry over from launch to launch), update fused and gradients materialised, with the training
    loop's monitor riding along; the status word stays 0.  Shapes: the benchmark's, PCD-1, config #2's, more row strips than
    8 (B = 512 at H = 128: 16 workgroups per group, two panels per K chain), a ragged number of W strips and 12 classes, a
    partial last row strip, H = 256 (B > 256 is not a shape the fused tail takes: those cases run the launch-per-phase tail in
    both modes).  The fused tail is an opt-in (mode 3: it is slower than the launch-per-phase tail)."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    seed, lr, sparse = 23, 0.05, 0.0005
    prev = ops.fused_chain_mode(-1)
    results = {}
    try:
        for mode in (3, 0):
            ops.fused_chain_mode(mode)
            for fuse in (True, False):
                W, b, d, U, c = synth_params(V, H, C, 1200 + V + H)
                P = P_of(ops, W, b, d, U, c)
                G = GradSet(V, H, C, ops.device)
                ws = ops.cd_workspace(P, B)
                cws = None
                sums = torch.zeros(2, dtype=torch.float64, device=ops.device)
                v_init, _, _, _, y_init = synth_inputs(B, V, H, C, 70)
                cv_o, cy_o = (v_init.copy(), y_init.copy()) if pcd else (None, None)
                cv, cy = (dev(ops, v_init.astype(np.uint8)), dev(ops, y_init)) if pcd else (None, None)
                for step in range(3):
                    v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 71 + step)
                    off, row0 = 4 * step, B * step
                    out = ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, seed, off, row0, cv, cy, fuse, lr, 1.0,
                                      float(B), sparse, ws)
                    if not fuse:
                        ops.update(P, G, lr, 1.0, float(B), sparse)
                    ph0, vk, yk, phk = oracle.rbm_cd_chain(v_bin, y1h, W, b, d, U, c, k, seed, off, row0, cv_o, cy_o)
                    g = oracle.rbm_two_phase_grad(v_bin, y1h, ph0, vk, yk, phk)
                    tag = f"mode {mode}, fuse {fuse}, step {step}"
                    assert_bitwise(out[4], phk, f"p(h_k) [{tag}]")
                    if not fuse:
                        assert_bitwise(G.gW, g[0], f"gW [{tag}]")
                    for got, want, nm in ((G.gb, g[1], "gb"), (G.gd, g[2], "gd"), (G.gU, g[3], "gU"), (G.gc, g[4], "gc")):
                        assert_bitwise(got, want, f"{nm} [{tag}]")
                    oracle.rbm_update(W, b, d, U, c, *g, lr, 1.0, float(B), sparse)
                    for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
                        assert_bitwise(got, want, f"{nm} [{tag}]")
                assert int(ws[5][512].item()) == 0
                results[(mode, fuse)] = [t.clone() for t in (P.W, P.b, P.d, P.U, P.c)]
        for fuse in (True, False):
            for x, y in zip(results[(3, fuse)], results[(0, fuse)]):
                assert torch.equal(x, y)
        # the monitor riding in the step: loss sums and parameters equal with and without the fused tail
        mon = {}
        for mode in (3, 0):
            ops.fused_chain_mode(mode)
            W, b, d, U, c = synth_params(V, H, C, 1300 + V + H)
            P = P_of(ops, W, b, d, U, c)
            G = GradSet(V, H, C, ops.device)
            sums = torch.tensor([1.5, 0.125], dtype=torch.float64, device=ops.device)
            for step in range(2):
                v_bin, _, _, labels, y1h = synth_inputs(B, V, H, C, 75 + step)
                ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, seed, 4 * step, 0, None, None, True, lr, 1.0, float(B),
                            sparse, monitor=(dev(ops, labels.astype(np.int32)), sums))
            ops.check_chain_status()
            mon[mode] = (sums.tolist(), P.W.clone(), P.c.clone())
        assert mon[3][0] == mon[0][0] and mon[3][0] != [1.5, 0.125]
        assert torch.equal(mon[3][1], mon[0][1]) and torch.equal(mon[3][2], mon[0][2])
    finally:
        ops.fused_chain_mode(prev)
