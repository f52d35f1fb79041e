"""Device-resident dataset pipeline.
CPU part: the oracle restatement (Philox shuffle keys, permutation, gather / bit unpacking) and DeviceDataset's host
logic through the oracle-backed double.  GPU part (-m gpu): the HIP entry points against the oracle, bit for bit."""
import numpy as np
import pytest

from helpers import golden

torch = pytest.importorskip("torch")


def _set(n=37, v=784, classes=3, seed=0):
    rs = np.random.RandomState(seed)
    x = rs.random_sample((n, v)).astype(np.float32)
    y = rs.randint(0, classes, n)
    return x, y


def test_oracle_shuffle_keys_and_permutation(oracle):
    keys = oracle.data_shuffle_keys(44, 3, 11)
    for i in range(11):   # known answer: word i&3 of the Philox block with counter (i>>2, 0, epoch lo, epoch hi)
        assert keys[i] == int(oracle.philox_raw([44, 0], [i >> 2, 0, 3, 0])[i & 3] >> 1)
    p0, p0b, p1 = oracle.data_shuffle_perm(44, 0, 1000), oracle.data_shuffle_perm(44, 0, 1000), oracle.data_shuffle_perm(44, 1, 1000)
    assert np.array_equal(np.sort(p0), np.arange(1000)) and np.array_equal(p0, p0b) and not np.array_equal(p0, p1)
    assert oracle.data_shuffle_perm(1, 0, 0).size == 0


def test_oracle_gather_unpacks_the_mnist_fixture(oracle):
    bits = golden("data_mnist01_t10k_bits.npz")["bits"]
    idx = np.array([5, 0, 77, 5])
    rows = oracle.data_gather(bits, idx, bits_visible=784, out_dtype=np.uint8)
    assert rows.shape == (4, 784) and np.array_equal(rows, np.unpackbits(bits[idx], axis=1)[:, :784])
    x, lab, one = oracle.data_gather(np.arange(12, dtype=np.float32).reshape(4, 3), [2, 1], labels=[0, 1, 2, 1], num_classes=3)
    assert np.array_equal(x, [[6, 7, 8], [3, 4, 5]]) and np.array_equal(lab, [2, 1]) and np.array_equal(one, [[0, 0, 1], [0, 1, 0]])


@pytest.mark.parametrize("storage,out", [("f32", "f32"), ("u8", "u8"), ("bits", "u8"), ("bits", "f32")])
@pytest.mark.parametrize("eg", [True, False], ids=["gather-per-epoch", "gather-per-batch"])
def test_device_dataset_host_logic(oracle, storage, out, eg):
    from oracle_backends import OracleDataOps
    from qbm_image_classification_amd.device_data import DeviceDataset

    x, y = _set()
    want = x if storage == "f32" else (x > 0.5).astype(np.float32)
    ds = DeviceDataset(x, y, batch_size=8, num_classes=3, storage=storage, out=out, seed=9, ops=OracleDataOps(), epoch_gather=eg)
    assert len(ds) == 5 and ds.epoch == 0
    seen = []
    for ep in range(2):
        perm = oracle.data_shuffle_perm(9, ep, 37)
        got = list(ds)
        assert [len(b) for b, _ in got] == [8, 8, 8, 8, 5] and ds.epoch == ep + 1
        xb = torch.cat([b for b, _ in got]).numpy().astype(np.float32)
        lb = torch.cat([l for _, l in got]).numpy()
        assert np.array_equal(xb, want[perm]) and np.array_equal(lb, y[perm]) and got[0][1].dtype == torch.int64
        seen.append(perm)
    assert not np.array_equal(seen[0], seen[1])
    # two ranks together deliver exactly the single-process batches of 16
    whole = DeviceDataset(x, y, batch_size=16, num_classes=3, storage=storage, out=out, seed=9, ops=OracleDataOps(), epoch_gather=eg)
    r0 = DeviceDataset(x, y, batch_size=8, num_classes=3, storage=storage, out=out, seed=9, rank=0, world=2, ops=OracleDataOps(), epoch_gather=eg)
    r1 = DeviceDataset(x, y, batch_size=8, num_classes=3, storage=storage, out=out, seed=9, rank=1, world=2, ops=OracleDataOps(), epoch_gather=eg)
    assert len(whole) == len(r0) == len(r1) == 3
    for (xw, lw, ow), (x0, l0, _), (x1, l1, _) in zip(whole.batches(0, True), r0.batches(0), r1.batches(0)):
        assert torch.equal(xw, torch.cat([x0, x1])) and torch.equal(lw, torch.cat([l0, l1]))
        assert torch.equal(ow.argmax(1).int(), lw)
    assert [len(b[0]) for b in r1.batches(0)] == [8, 8, 0]      # 37 = 16 + 16 + 5: the last global batch holds 5 rows
    # no shuffle, drop_last
    plain = DeviceDataset(x, y, batch_size=8, num_classes=3, storage=storage, out=out, shuffle=False, drop_last=True, ops=OracleDataOps(), epoch_gather=eg)
    assert len(plain) == 4 and np.array_equal(next(iter(plain))[1].numpy(), y[:8])
    with pytest.raises(ValueError):
        DeviceDataset(x, y, batch_size=8, num_classes=3, storage="f32", out="u8", ops=OracleDataOps(), epoch_gather=eg)


# ---- GPU: HIP entry points vs the oracle ---------------------------------------------------------------------
@pytest.fixture(scope="module")
def dops():
    from qbm_image_classification_amd.device_data import HipDataOps

    return HipDataOps("cuda:0")


@pytest.mark.gpu
def test_hip_shuffle_pack_binarise_bitwise(dops, oracle):
    for n in (1, 5, 37, 2115, 60000):
        assert np.array_equal(dops.shuffle_keys(44, 7, n).cpu().numpy(), oracle.data_shuffle_keys(44, 7, n))
        assert np.array_equal(dops.permutation(44, 7, n).cpu().numpy(), oracle.data_shuffle_perm(44, 7, n))
    x, _ = _set(50, 13)
    xb = dops.binarise(torch.from_numpy(x), 0.5)
    assert np.array_equal(xb.cpu().numpy(), (x > 0.5).astype(np.uint8))
    assert np.array_equal(dops.pack_bits(xb).cpu().numpy(), np.packbits((x > 0.5).astype(np.uint8), axis=1))
    f = golden("data_mnist01_t10k_bits.npz")
    full = np.unpackbits(f["bits"], axis=1)[:, :784]
    assert np.array_equal(dops.pack_bits(torch.from_numpy(full)).cpu().numpy(), f["bits"])   # 98 B per image


@pytest.mark.gpu
@pytest.mark.parametrize("V", [784, 13])
def test_hip_gather_all_kinds(dops, oracle, V):
    x, y = _set(300, V, classes=10, seed=3)
    xb = (x > 0.5).astype(np.uint8)
    idx = np.random.RandomState(1).randint(0, 300, 129)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dops.device)
    lab = t(y.astype(np.int32))
    srcs = {0: (t(x), x, None), 1: (t(xb), xb, None), 2: (t(np.packbits(xb, axis=1)), np.packbits(xb, axis=1), V)}
    for sk, (dev_src, host_src, bv) in srcs.items():
        for dk, dt in ((0, np.float32), (1, np.uint8)):
            if sk == 0 and dk == 1:
                continue    # fp32 greyscale -> uint8 is defined for binary data only
            got, gl, go = dops.gather(dev_src, sk, t(idx.astype(np.int32)), V, dk, lab, 10, True)
            wx, wl, wo = oracle.data_gather(host_src, idx, y, 10, bv, dt)
            assert got.dtype == (torch.float32 if dk == 0 else torch.uint8)
            assert np.array_equal(got.cpu().numpy(), wx) and np.array_equal(gl.cpu().numpy(), wl) and np.array_equal(go.cpu().numpy(), wo)
    got, gl, go = dops.gather(srcs[1][0], 1, t(idx[:3].astype(np.int32)), V, 1)
    assert gl is None and go is None and np.array_equal(got.cpu().numpy(), xb[idx[:3]])


@pytest.mark.gpu
def test_training_from_device_dataset_equals_manual_batches(dops):
    """ClassificationRBM.train_rbm fed by DeviceDataset (bit-packed storage, uint8 minibatches) ends with the same
    parameters, bit for bit, as the same batches gathered on the host and passed in as fp32."""
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.device_data import DeviceDataset

    x, y = _set(70, 784, classes=2, seed=5)
    xb = (x > 0.5).astype(np.float32)

    def model():
        return ClassificationRBM(784, 32, 1, num_classes=2, learning_rate=0.05, seed=1, device="cuda:0")

    a = model()
    ds = DeviceDataset(x, y, batch_size=16, num_classes=2, storage="bits", seed=3)
    assert ds.bytes_resident() == 70 * 98 + 70 * 4
    a.train_rbm(ds, epochs=2, method="hybrid")
    b = model()
    manual = []
    for ep in range(2):
        perm = ds.ops.permutation(3, ep, 70).cpu().numpy()
        manual.append([(torch.from_numpy(xb[perm[i:i + 16]]), torch.from_numpy(y[perm[i:i + 16]])) for i in range(0, 70, 16)])

    class Loader:
        batch_size = 16   # like torch's DataLoader: the nominal minibatch, by which the Philox row counter advances per step

        def __init__(self):
            self.ep = 0

        def __len__(self):
            return 5

        def __iter__(self):
            self.ep += 1
            return iter(manual[self.ep - 1])

    b.train_rbm(Loader(), epochs=2, method="hybrid")
    for nm in ("weights", "visible_bias", "hidden_bias", "class_weights", "class_bias"):
        assert torch.equal(getattr(a, nm), getattr(b, nm)), nm


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["bits", "f32"])
def test_discriminative_epochs_replayed_from_one_graph_equal_eager_epochs(dops, storage, monkeypatch):
    """train_rbm(method='discriminative') from a DeviceDataset captures an epoch's launches in ONE hipGraph at the second
    epoch and replays it from then on (the dataset keeps its epoch buffers, the rule draws no random numbers, the loss sums
    are device memory): parameters, loss and NLL per epoch equal the eager loop's bit for bit (QBM_EPOCH_GRAPH=0), over four
    epochs with a short last minibatch, bit-packed and fp32 storage."""
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.device_data import DeviceDataset

    x, y = _set(150, 784, classes=2, seed=9)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("QBM_EPOCH_GRAPH", flag)
        m = ClassificationRBM(784, 64, 1, num_classes=2, learning_rate=0.05, seed=2, device="cuda:0")
        ds = DeviceDataset(x, y, batch_size=32, num_classes=2, storage=storage, seed=4)
        loss, _, nll = m.train_rbm(ds, epochs=4, method="discriminative")
        outs.append((loss, nll, [getattr(m, nm).clone() for nm in ("weights", "visible_bias", "hidden_bias", "class_weights", "class_bias")]))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1]
    assert len(set(outs[0][0])) == 4                                   # the epochs differ: the replay trained on
    for a, b in zip(outs[0][2], outs[1][2]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["bits", "f32"])
def test_generative_epochs_replayed_from_one_graph_equal_eager_epochs(dops, storage, monkeypatch):
    """train_rbm(method='generative'), CD-1, from a DeviceDataset: the epoch's launches are captured with draw ids and Philox
    rows RELATIVE to the epoch, two device words (qbm_rbm_cd_workspace.draw_base) carry the epoch's base and are moved between
    replays -- the draws, the monitor's loss sums and the parameters equal the eager loop's (QBM_EPOCH_GRAPH=0) bit for bit over
    four epochs with a short last minibatch; and a second train_rbm call on the same object goes on from the captured graph."""
    from qbm_image_classification_amd.classification_rbm import ClassificationRBM
    from qbm_image_classification_amd.device_data import DeviceDataset

    x, y = _set(150, 784, classes=2, seed=11)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("QBM_EPOCH_GRAPH", flag)
        m = ClassificationRBM(784, 64, 1, num_classes=2, learning_rate=0.05, seed=2, device="cuda:0")
        ds = DeviceDataset(x, y, batch_size=32, num_classes=2, storage=storage, seed=4)
        loss, _, nll = m.train_rbm(ds, epochs=3, method="generative")
        loss2, _, nll2 = m.train_rbm(ds, epochs=1, method="generative")
        outs.append((loss + loss2, nll + nll2, m._draw_offset, m._rows_seen,
                     [getattr(m, nm).clone() for nm in ("weights", "visible_bias", "hidden_bias", "class_weights", "class_bias")]))
        assert (m._epoch_graph is not None) == (flag == "1")
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and outs[0][2:4] == outs[1][2:4]
    assert len(set(outs[0][0])) == 4
    for a, b in zip(outs[0][4], outs[1][4]):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_device_side_draw_counters_equal_host_side_offsets(dops):
    """qbm_rbm_cd_workspace.draw_base: a CD-2 / PCD step (launch per phase) with (offset, row0) = (5, 7) in the arguments and
    (1000, 2^32 - 3) in the two device words == the same step with (1005, 4) in the arguments: every sampled bit and parameter."""
    from qbm_image_classification_amd.rbm_ops import GradSet, HipRbmOps, ParamSet

    ops = HipRbmOps("cuda:0")
    rs = np.random.RandomState(3)
    V, H, C, B = 784, 128, 2, 64
    mk = lambda: ParamSet(*(torch.from_numpy(a).cuda() for a in ((rs2.standard_normal((V, H)) * 0.1).astype(np.float32), np.zeros(V, np.float32),
                                                                 np.zeros(H, np.float32), (rs2.standard_normal((C, H)) * 0.1).astype(np.float32),
                                                                 np.zeros(C, np.float32))))
    v0 = torch.from_numpy((rs.random_sample((B, V)) < 0.3).astype(np.uint8)).cuda()
    y0 = torch.nn.functional.one_hot(torch.from_numpy(rs.randint(0, C, B)), C).float().cuda()
    res = []
    for base, off, row0 in ((torch.tensor([1000, 2 ** 32 - 3], dtype=torch.int64).cuda(), 5, 7), (None, 1005, 4)):
        rs2 = np.random.RandomState(8)
        P, G = mk(), GradSet(V, H, C, ops.device)
        cv, cy = v0.roll(1, 0).clone(), y0.roll(1, 0).clone()
        ops.cd_step(P, G, v0, y0, 2, 9, off, row0, cv, cy, True, 0.05, 1.0, float(B), 0.0, draw_base=base)
        res.append([t.clone() for t in (P.W, P.d, P.U, cv, cy)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
