"""GPU: the fail-safe of the one-launch Gibbs chain (include/qbm_hip.h: qbm_rbm_cd_workspace.sync) and the device-side row
count of the data-parallel update (qbm_rbm_update_dev).  No attempt is made to provoke a real failure of the chain kernel:
the status word is pre-set, which is exactly what the launches behind a failed chain would find."""
import warnings

import numpy as np
import pytest

from helpers import synth_inputs, synth_params
from test_gpu_parity import P_of, assert_bitwise, dev

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture()
def ops():
    from qbm_image_classification_amd.rbm_ops import HipRbmOps

    o = HipRbmOps("cuda:0")
    prev = o.fused_chain_mode(2)      # the fused kernel for every shape it takes
    yield o
    o.fused_chain_mode(prev)


def _oracle_step(oracle, W, b, d, U, c, v, y, k, seed, off, row0, lr, B, sparse):
    ph0, vk, yk, phk = oracle.rbm_cd_chain(v, y, W, b, d, U, c, k, seed, off, row0)
    g = oracle.rbm_two_phase_grad(v, y, ph0, vk, yk, phk)
    oracle.rbm_update(W, b, d, U, c, *g, lr, 1.0, float(B), sparse)


@pytest.mark.parametrize("fuse", [True, False])
def test_failed_chain_never_reaches_parameters_and_is_replayed(ops, oracle, fuse):
    """Three CD-2 steps on the workspaces cd_step keeps, the status word set before the SECOND one (as if its chain had
    given up; word 513 = that step's draw offset, as the kernel records it).  The second and third steps must change
    nothing on the device; check_chain_status then repeats both on the launch-per-phase path: parameters equal the oracle's
    after three steps, bit for bit, the process is on mode 0, the buffer is clean."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    V, H, C, B, k = 784, 128, 2, 64, 2
    W, b, d, U, c = synth_params(V, H, C, 77)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    seed, lr, sparse = 5, 0.05, 0.001
    data = [synth_inputs(B, V, H, C, 80 + s) for s in range(3)]

    def step(s):
        v8, y = dev(ops, data[s][0].astype(np.uint8)), dev(ops, data[s][4])
        off, row0 = 7 * s, B * s

        def run():
            ops.cd_step(P, G, v8, y, k, seed, off, row0, None, None, fuse, lr, 1.0, float(B), sparse, replay=run)
            if not fuse:
                ops.update(P, G, lr, 1.0, float(B), sparse, guard_chain=True)
        run()

    step(0)
    _oracle_step(oracle, W, b, d, U, c, data[0][0], data[0][4], k, seed, 0, 0, lr, B, sparse)
    assert_bitwise(P.W, W, "W after the first (healthy) step")
    sync = ops._sync
    assert sync is not None and int(sync[512]) == 0
    sync[512:514] = torch.tensor([1, 7], dtype=torch.int32, device=sync.device)   # "step 1 (offset 7): a barrier timed out"
    step(1)
    step(2)
    torch.cuda.synchronize()
    for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
        assert_bitwise(got, want, f"{nm} untouched behind a chain that gave up")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ops.check_chain_status()
    assert any("gave up" in str(w.message) and "Repeating 2 step" in str(w.message) for w in rec), [str(w.message) for w in rec]
    assert ops.fused_chain_mode() == 0 and not sync.any() and ops._pending == []
    for s in (1, 2):
        _oracle_step(oracle, W, b, d, U, c, data[s][0], data[s][4], k, seed, 7 * s, B * s, lr, B, sparse)
    for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
        assert_bitwise(got, want, f"{nm} after the replay")
    ops.check_chain_status()                                                      # nothing pending, nothing set: silent


@pytest.mark.parametrize("path", ["fuse", "with_count"])
def test_step_of_an_ineligible_shape_behind_a_failed_chain_is_a_noop_and_replayed_in_order(oracle, path):
    """in the DEFAULT mode (1: the one-launch chain from k = 2 on) an epoch of full minibatches of 64
    and a short last one of 36 -- a shape the chain kernel does not take (PneumoniaMNIST: 4708 % 64).  The status word is set
    before the second step: the full minibatch behind it AND the short last one must change nothing (the short step runs one
    launch per phase, every storing launch reading the word); check_chain_status then repeats steps 1, 2, 3 in order.
    Parameters equal the oracle's after four steps bit for bit -- with the update fused into the gradient launch and on the
    data-parallel form (row count -> update launch)."""
    from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps

    ops = HipRbmOps("cuda:0")
    prev = ops.fused_chain_mode(1)
    try:
        V, H, C, k = 784, 128, 2, 2
        W, b, d, U, c = synth_params(V, H, C, 71)
        P = P_of(ops, W, b, d, U, c)
        G = GradSet(V, H, C, ops.device)
        seed, lr, sparse = 5, 0.05, 0.001
        sizes = [64, 64, 64, 36]
        data = [synth_inputs(B, V, H, C, 60 + s) for s, B in enumerate(sizes)]
        fuse = path == "fuse"

        def step(s):
            B = sizes[s]
            v8, y = dev(ops, data[s][0].astype(np.uint8)), dev(ops, data[s][4])
            off, row0 = 7 * s, 64 * s

            def run():
                ops.cd_step(P, G, v8, y, k, seed, off, row0, None, None, fuse, lr, 1.0, float(B), sparse, with_count=not fuse, replay=run)
                if not fuse:
                    ops.update(P, G, lr, 1.0, 0.0, sparse, count_dev=G.count)     # the data-parallel form: -inf rows = no-op
            run()

        step(0)
        _oracle_step(oracle, W, b, d, U, c, data[0][0], data[0][4], k, seed, 0, 0, lr, 64, sparse)
        assert_bitwise(P.W, W, "W after the first (healthy) step")
        sync = ops._sync
        sync[512:514] = torch.tensor([1, 7], dtype=torch.int32, device=sync.device)   # "step 1 (offset 7): a barrier timed out"
        for s in (1, 2, 3):
            step(s)
        torch.cuda.synchronize()
        for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
            assert_bitwise(got, want, f"{nm} untouched behind a chain that gave up (the short last minibatch included)")
        if not fuse:
            assert float(G.count) == float("-inf")
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            ops.check_chain_status()
        assert any("Repeating 3 step" in str(w.message) for w in rec), [str(w.message) for w in rec]
        for s in (1, 2, 3):
            _oracle_step(oracle, W, b, d, U, c, data[s][0], data[s][4], k, seed, 7 * s, 64 * s, lr, sizes[s], sparse)
        for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
            assert_bitwise(got, want, f"{nm} after the replay")
    finally:
        ops.fused_chain_mode(prev)


def test_hybrid_discriminative_steps_behind_a_failed_chain_are_repeated_in_order(ops, oracle):
    """hybrid training alternates discriminative and CD steps on the same parameters.  With the status word
    set before the second CD step, the discriminative steps behind it must store nothing either (parameters, loss sums) and
    be repeated IN ORDER with the CD steps: two objects -- one disturbed, one on the launch-per-phase path from the start --
    end with the same bits."""
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.device_data import DeviceDataset

    rs = np.random.RandomState(3)
    V, H, C, N, B = 784, 128, 2, 4 * 64, 64
    x = (rs.random_sample((N, V)) < 0.25).astype(np.uint8)
    y = rs.randint(0, C, N)
    outs = []
    for disturbed in (True, False):
        ops.fused_chain_mode(2 if disturbed else 0)
        m = ClassificationRBM(V, H, 2, num_classes=C, learning_rate=0.05, seed=2, device="cuda:0")
        ds = DeviceDataset(x, y, batch_size=B, num_classes=C, storage="u8", seed=4)
        if disturbed:
            orig, calls = m.contrastive_divergence_training, []

            def spy(*a, **k):
                if len(calls) == 1:      # in front of the second CD step: "its chain gave up" (word 513 = its draw offset)
                    sync = m.ops._sync_buffer()
                    sync[512:514] = torch.tensor([1, m._draw_offset], dtype=torch.int32, device=sync.device)
                calls.append(1)
                return orig(*a, **k)

            m.contrastive_divergence_training = spy
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            loss, _, nll = m.train_rbm(ds, epochs=1, method="hybrid", generative_factor=0.1)
        if disturbed:
            # 4 minibatches: CD steps 2..4 and the discriminative steps 3, 4 (step 2's ran in front of the lost CD step)
            assert any("Repeating 5 step" in str(w.message) for w in rec), [str(w.message) for w in rec]
        outs.append((loss, nll, [getattr(m, nm).clone() for nm in ("weights", "visible_bias", "hidden_bias", "class_weights", "class_bias")]))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


def test_caller_owned_workspace_with_status_set(ops):
    """A caller-owned workspace whose status word is set: the chain and the gradient launch do nothing (the row-count slot
    reads -inf, which turns a data-parallel update into a no-op on every rank), qbm_rbm_update_dev honours both guards,
    check_chain_status(ws) raises."""
    from qbm_image_classification_amd._lib import QbmHipError
    from qbm_image_classification_amd.rbm_ops import GradSet

    V, H, C, B, k = 784, 128, 2, 64, 2
    W, b, d, U, c = synth_params(V, H, C, 78)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    G.flat.fill_(3.0)
    v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 90)
    ws = ops.cd_workspace(P, B)
    ws[5][512] = 2
    for t in ws[:5]:
        t.fill_(0)
    ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, 1, 0, 0, None, None, True, 0.05, 1.0, float(B), 0.0, ws,
                with_count=True)
    torch.cuda.synchronize()
    assert_bitwise(P.W, W, "W")
    assert not ws[0].any() and not ws[2].any()                                   # the chain did not run
    assert (G.flat[:-1] == 3.0).all() and float(G.count) == float("-inf")
    ops.update(P, G, 0.05, 1.0, 0.0, 0.0, count_dev=G.count)                      # -inf rows: no-op
    assert_bitwise(P.W, W, "W after update_dev(-inf)")
    ops._sync = ws[5]
    ops.update(P, G, 0.05, 1.0, float(B), 0.0, guard_chain=True)                  # status set: no-op
    ops._sync = None
    assert_bitwise(P.W, W, "W after the guarded update")
    with pytest.raises(QbmHipError, match="gave up"):
        ops.check_chain_status(ws)


@pytest.mark.parametrize("V,H,C,B", [(784, 512, 10, 256), (13, 9, 3, 5)])
def test_update_with_device_row_count_equals_scalar(ops, V, H, C, B):
    """qbm_rbm_update_dev with the batch size read from the row-count slot the gradient launch wrote == qbm_rbm_update with
    the scalar, bit for bit (the data-parallel step has no host round trip)."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    W, b, d, U, c = synth_params(V, H, C, 79)
    v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 91)
    outs = []
    for on_device in (False, True):
        P = P_of(ops, W, b, d, U, c)
        G = GradSet(V, H, C, ops.device)
        ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), 1, 3, 0, 0, None, None, False, 0.05, 1.0, float(B), 0.001,
                    with_count=on_device)
        if on_device:
            assert float(G.count) == float(B)
            ops.update(P, G, 0.05, 1.0, 0.0, 0.001, count_dev=G.count)
        else:
            assert float(G.count) == 0.0
            ops.update(P, G, 0.05, 1.0, float(B), 0.001)
        outs.append([t.clone() for t in (P.W, P.b, P.d, P.U, P.c)])
    for a, bb in zip(*outs):
        assert torch.equal(a, bb)
    assert not torch.equal(outs[0][0], dev(ops, W))
    # the discriminative step writes its row count too
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    ops.disc_step(P, G, dev(ops, v_bin), torch.zeros(B, dtype=torch.int32), False, False, 0.05, 1.0, float(B), 0.0, with_count=True)
    assert float(G.count) == float(B)


def test_short_last_minibatch_keeps_the_status_buffer(ops):
    """the workspace cd_step keeps is re-made when the minibatch shape changes (B = 256 -> 96 at the end of
    an epoch); the state buffer of the fused chain kernel must outlive that -- it is ONE buffer per HipRbmOps."""
    from qbm_image_classification_amd.rbm_ops import GradSet

    V, H, C, k = 784, 128, 2, 2
    W, b, d, U, c = synth_params(V, H, C, 81)
    P = P_of(ops, W, b, d, U, c)
    G = GradSet(V, H, C, ops.device)
    for B in (256, 96, 256):
        v_bin, _, _, _, y1h = synth_inputs(B, V, H, C, 92)
        ops.cd_step(P, G, dev(ops, v_bin.astype(np.uint8)), dev(ops, y1h), k, 3, 0, 0, None, None, True, 0.05, 1.0, float(B), 0.0)
    assert len(ops._cd_wss) == 2 and len({id(w[5]) for w in ops._cd_wss.values()}) == 1
    assert len(ops._pending) == 3
    ops.check_chain_status()
    assert ops._pending == [] and int(ops._sync[512]) == 0


@pytest.mark.parametrize("kind,fuse", [(2, True), (1, True), (2, False)])
def test_fused_tail_that_gives_up_is_finished_and_later_steps_repeated(oracle, kind, fuse):
    """The fused tail of a CD-1 step (k_hgrad) with ONE workgroup made to misbehave in the second of four steps
    (qbm_test_hgrad_fault) -- kind 2: it gives up in front of the barrier (its partners complete their share of the update:
    the step is PARTLY applied); kind 1: it leaves at its start as if it had not been resident in time (its partners' bounded
    barrier really times out: none of that column group's items run, and its own hidden phase is missing).  Steps 3 and 4 must
    change nothing; check_chain_status then finishes step 2 from its workspace (with the update inside the launch) or repeats
    it (gradients materialised: the parameters were never touched) and repeats steps 3 and 4: parameters equal the oracle's
    after four undisturbed steps, bit for bit."""
    from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps

    ops = HipRbmOps("cuda:0")
    prev = ops.fused_chain_mode(3)      # the fused tail is an opt-in
    try:
        V, H, C, B, k = 784, 512, 10, 256, 1
        W, b, d, U, c = synth_params(V, H, C, 177)
        P = P_of(ops, W, b, d, U, c)
        G = GradSet(V, H, C, ops.device)
        seed, lr, sparse = 5, 0.05, 0.001
        data = [synth_inputs(B, V, H, C, 180 + s) for s in range(4)]

        def step(s):
            v8, y = dev(ops, data[s][0].astype(np.uint8)), dev(ops, data[s][4])
            off, row0 = 4 * s, B * s

            def run():
                ops.cd_step(P, G, v8, y, k, seed, off, row0, None, None, fuse, lr, 1.0, float(B), sparse, replay=run)
                if not fuse:
                    ops.update(P, G, lr, 1.0, float(B), sparse, guard_chain=True)
            run()

        step(0)
        _oracle_step(oracle, W, b, d, U, c, data[0][0], data[0][4], k, seed, 0, 0, lr, B, sparse)
        assert_bitwise(P.W, W, "W after the first (healthy) step")
        ops.lib.qbm_test_hgrad_fault(77, kind)                                # workgroup 77 of the NEXT fused tail
        step(1)
        torch.cuda.synchronize()
        st = ops._sync[512:516].tolist()
        assert st[0] == 3 and st[1] == 4 and st[3] == 2, st                   # barrier, draw offset of step 2, round 2
        snap = [t.clone() for t in (P.W, P.b, P.d, P.U, P.c)]
        if fuse and kind == 2:
            assert not torch.equal(P.W, dev(ops, W)), "the healthy column groups have applied their update"
        if not fuse:
            assert_bitwise(P.W, W, "W untouched (gradients materialised, guarded update)")
        step(2)
        step(3)
        torch.cuda.synchronize()
        for a, bb in zip(snap, (P.W, P.b, P.d, P.U, P.c)):
            assert torch.equal(a, bb), "steps behind a fused tail that gave up must not store anything"
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            ops.check_chain_status()
        msg = [str(w.message) for w in rec]
        assert any("gave up" in m and ("Finished that step" in m) == fuse for m in msg), msg
        assert ops.fused_chain_mode() == 0 and not ops._sync.any() and ops._pending == []
        for s in (1, 2, 3):
            _oracle_step(oracle, W, b, d, U, c, data[s][0], data[s][4], k, seed, 4 * s, B * s, lr, B, sparse)
        for got, want, nm in ((P.W, W, "W"), (P.b, b, "b"), (P.d, d, "d"), (P.U, U, "U"), (P.c, c, "c")):
            assert_bitwise(got, want, f"{nm} after finishing / repeating")
    finally:
        ops.fused_chain_mode(prev)
