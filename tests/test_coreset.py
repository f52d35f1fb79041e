"""Greedy k-center coreset (src/coreset.py).
CPU: the oracle restatement against golden vectors produced by the imported reference (tools/make_golden.py
gen_coreset).  GPU (-m gpu): the HIP picks against the oracle, index for index, and the drop-in function against
the same goldens."""
import numpy as np
import pytest

from helpers import golden

torch = pytest.importorskip("torch")


def test_oracle_reproduces_reference_selections(oracle):
    g = golden("coreset_small.npz")
    assert np.array_equal(oracle.k_center_greedy(g["emb"], int(g["budget"]), int(g["random_seed"])), g["idx"])
    m = golden("coreset_mid.npz")
    a = oracle.k_center_greedy(m["emb"], 75, 11)
    assert len(a) == 75 and np.array_equal(a, m["idx_budget75_seed11"])
    b = oracle.k_center_greedy(m["emb"], 40, 5, index=m["index"], already_selected=[3, 17, 599])
    assert len(b) == 40 and np.array_equal(b, m["idx_budget40_seed5_given"])


def test_oracle_picks_edge_cases(oracle):
    emb = np.array([[0, 0], [3, 4], [0, 1], [10, 0], [0, 0]], np.float32)
    picks, sel = oracle.kcenter_picks(emb, [0], 3)
    # first pick re-selects the seed (+inf at the selected entries, first index wins), then the farthest points
    assert picks.tolist() == [0, 3, 1] and sel.tolist() == [1, 1, 0, 1, 0]
    # budget >= N: the reference clamps budget+1 to N and its first pick re-selects the seed -> N-1 points (mirrored)
    assert len(oracle.k_center_greedy(emb, 10, random_seed=0)) == 4
    one = oracle.k_center_greedy(emb, 0, random_seed=1)                                           # budget 0 -> the seed only
    assert len(one) == 1


def test_random_downsampling_matches_reference():
    from qbm_image_classification_amd import coreset

    m = golden("coreset_mid.npz")
    xs, ys = coreset.random_downsampling(np.arange(50) * 2, np.arange(50), 7, seed=42)
    assert np.array_equal(xs, m["random_down_x"]) and np.array_equal(ys, m["random_down_y"])
    with pytest.raises(ValueError):
        coreset.random_downsampling(np.arange(3), np.arange(3), 5)
    with pytest.raises(NotImplementedError):
        coreset.k_center_greedy(np.zeros((4, 2), np.float32), 2, metric=lambda a, b: a)
    with pytest.raises(TypeError):
        coreset.k_center_greedy([[0.0, 1.0]], 1)


@pytest.mark.gpu
def test_hip_kcenter_matches_oracle_and_reference(oracle):
    from qbm_image_classification_amd import coreset

    g, m = golden("coreset_small.npz"), golden("coreset_mid.npz")
    assert np.array_equal(coreset.k_center_greedy(torch.from_numpy(g["emb"]), int(g["budget"]), random_seed=int(g["random_seed"])), g["idx"])
    assert np.array_equal(coreset.k_center_greedy(m["emb"], 75, random_seed=11), m["idx_budget75_seed11"])
    assert np.array_equal(coreset.k_center_greedy(m["emb"], 40, random_seed=5, index=m["index"], already_selected=[3, 17, 599]),
                          m["idx_budget40_seed5_given"])
    # pick ORDER and mask against the oracle on shapes the goldens do not cover (more rows than one launch's
    # workgroups x threads in one stride, wide embeddings, duplicate points = exact distance ties)
    rs = np.random.RandomState(4)
    for n, d, seeds, picks in ((5000, 256, [17], 300), (300, 7, [0, 299, 5], 120), (263, 1, [2], 50), (1000, 8, [0, 999, 5], 400),
                               (257, 4, [256], 257), (90, 64, [3], 40)):
        emb = rs.standard_normal((n, d)).astype(np.float32)
        emb[n // 2:n // 2 + 20] = emb[:20]                      # duplicates
        want_p, want_s = oracle.kcenter_picks(emb, seeds, picks)
        # both device paths: all picks in ONE launch (persistent workgroups, rows in LDS, one granule per pick: shapes with
        # D a multiple of 4) and one launch per pick
        for fused in ((True, False) if d % 4 == 0 else (False,)):
            got_p, got_s = coreset.kcenter_picks(emb, seeds, picks, fused=fused)
            assert np.array_equal(got_p.cpu().numpy(), want_p) and np.array_equal(got_s.cpu().numpy(), want_s), (n, d, fused)
        got_p, got_s = coreset.kcenter_picks(emb, seeds, picks)    # the default: whichever takes the shape
        assert np.array_equal(got_p.cpu().numpy(), want_p) and np.array_equal(got_s.cpu().numpy(), want_s)


@pytest.mark.gpu
def test_hip_kcenter_full_size_properties():
    """Size-independent properties at the size config #5 uses (MNIST 0/1 train split, 256-d hidden means): every
    pick is new after the first, the selection is exactly `budget` points, and the covering radius never grows."""
    from qbm_image_classification_amd import coreset

    rs = np.random.RandomState(0)
    emb = (rs.random_sample((12665, 256)) < 0.3).astype(np.float32) * rs.random_sample((12665, 256)).astype(np.float32)
    picks, sel = coreset.kcenter_picks(emb, [123], 501, fused=True)      # 256 persistent workgroups, 50 rows each in LDS
    picks1, sel1 = coreset.kcenter_picks(emb, [123], 501, fused=False)   # one launch per pick
    assert torch.equal(picks, picks1) and torch.equal(sel, sel1)
    p = picks.cpu().numpy()
    assert p[0] == 123 and len(set(p.tolist())) == 501 and int(sel.sum().item()) == 501
    e = torch.from_numpy(emb).cuda()
    radius = []
    for k in (50, 200, 501):
        radius.append(torch.cdist(e, e[torch.from_numpy(p[:k]).long().cuda()]).min(1).values.max().item())
    assert radius[0] >= radius[1] >= radius[2] > 0
    idx = coreset.k_center_greedy(emb, 500, random_seed=1)
    assert len(idx) == 500 and np.all(np.diff(idx) > 0)
