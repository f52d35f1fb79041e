"""CPU: host logic of the Disc_QBM mirror against golden vectors from the reference's own numpy methods
(faster_dqbm.py run with stub D-Wave modules, tools/make_golden.py).  Sampling is replaced by the fixed
sample sets the golden update was generated with -- the same mock the generating script applied to the
reference -- through an oracle-backed test double of the HIP ops."""
import numpy as np
import pytest

from helpers import golden
from oracle_backends import OracleQuboOps


def make(g, restricted=False, ops=None):
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    np.random.seed(int(g["pre_seed"]))   # W_hh is drawn before the ctor reseeds (faster_dqbm.py:82-83)
    return Disc_QBM(dim_input=int(g["D"]), num_classes=2, n_hidden_nodes=int(g["H"]), seed=int(g["seed"]), solver="HIP",
                    restricted=restricted, sample_count=int(g["R"]), anneal_steps=10, beta_eff=float(g["beta_eff"]),
                    ops=ops)


def test_ctor_parameter_order_matches_reference():
    g = golden("qbm_small.npz")
    q = make(g)
    for name, attr in (("W_vh", q.weights_all_visible_to_hidden), ("W_vo", q.weights_clamped_visible_to_output),
                       ("W_oo", q.weights_output_output), ("b_h", q.biases_hidden), ("b_o", q.biases_output),
                       ("W_hh", q.weights_hidden_hidden)):
        assert np.array_equal(attr, g[name]), name
    assert q.weight_objects[0] is q.weights_all_visible_to_hidden and len(q.weight_objects) == 6


def test_create_qubo_matrix_matches_reference():
    g = golden("qbm_small.npz")
    q = make(g)
    for i in range(len(g["X"])):
        assert np.allclose(q.create_qubo_matrix_from(g["X"][i], g["Y"][i]), g["Q_clamped"][i], rtol=0, atol=1e-15)
        assert np.allclose(q.create_qubo_matrix_from(g["X"][i]), g["Q_unclamped"][i], rtol=0, atol=1e-15)


def test_unknown_solver_raises_like_reference():
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    with pytest.raises(Exception, match="No valid solver"):
        Disc_QBM(dim_input=4, num_classes=2, solver="Advantage_system4.1")


def test_statistics_and_update_match_reference():
    g = golden("qbm_small.npz")
    H = int(g["H"])
    fixed = {H: g["S_clamped"], H + 1: g["S_unclamped"]}
    q = make(g, ops=OracleQuboOps(fixed))
    X, Y = g["X"], g["Y"]
    to_dicts = lambda S: [[dict(enumerate(map(float, S[i, r]))) for r in range(S.shape[1])] for i in range(S.shape[0])]
    for tag, S, labels in (("clamped", g["S_clamped"], Y), ("unclamped", g["S_unclamped"], None)):
        out = q.get_average_configuration_batch(to_dicts(S), X, labels)   # the reference's list[list[dict]] input
        for nme, arr in zip(["b_h", "b_o", "W_vh", "W_vo", "W_oo", "W_hh"], out):
            assert np.max(np.abs(np.asarray(arr) - g[f"avg_{tag}_{nme}"])) <= 1e-12, (tag, nme)
    err_bo, nll = q.train_for_one_iteration(X, Y, 0.1)
    assert np.allclose(err_bo, g["upd_err_b_o"], rtol=0, atol=1e-15)
    assert abs(nll - float(g["upd_nll"])) <= 1e-6
    for nme, attr in (("W_vh", q.weights_all_visible_to_hidden), ("W_vo", q.weights_clamped_visible_to_output),
                      ("W_oo", q.weights_output_output), ("b_h", q.biases_hidden), ("b_o", q.biases_output),
                      ("W_hh", q.weights_hidden_hidden)):
        assert np.max(np.abs(attr - g[f"upd_{nme}"])) <= 1e-12, nme
    assert np.array_equal(q.weights_hidden_hidden, g["W_hh"])            # Q8: hidden-hidden weights never learn


def test_predict_postprocessing_matches_reference():
    g = golden("qbm_small.npz")
    H = int(g["H"])
    q = make(g, ops=OracleQuboOps({H + 1: g["S_unclamped"][:1]}))
    pred, outs = q.predict(g["X"][0])
    assert pred == int(g["pred0"]) and np.array_equal(outs, g["pred0_outs"])
    half = np.zeros((1, int(g["R"]), H + 1), np.uint8)
    half[0, 1::2, 0] = 1                                                  # mean exactly 0.5 -> np.round -> 0
    q2 = make(g, ops=OracleQuboOps({H + 1: half}))
    assert q2.predict(g["X"][0])[0] == int(g["pred_half"]) == 0
    assert q2.predict_batch(g["X"][:1])[0] == 0


def test_restricted_model_works_and_samples_have_reference_format():
    g = golden("qbm_small.npz")
    q = make(g, restricted=True, ops=OracleQuboOps())
    s = q.get_samples(g["X"][0])
    assert len(s) == int(g["R"]) and set(s[0].keys()) == set(range(int(g["H"]) + 1)) and set(s[0].values()) <= {0.0, 1.0}
    sc = q.get_samples_batch(g["X"][:2], g["Y"][:2])
    assert len(sc) == 2 and set(sc[0][0].keys()) == set(range(int(g["H"])))
    q.train_for_one_iteration(g["X"], g["Y"], 0.05)                       # Q9: the reference crashes here
    assert q.weights_hidden_hidden is None


def test_checkpoint_roundtrip(tmp_path):
    g = golden("qbm_small.npz")
    q = make(g, ops=OracleQuboOps())
    q.save_weights("ck", str(tmp_path))
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    q2 = Disc_QBM(dim_input=int(g["D"]), num_classes=2, n_hidden_nodes=int(g["H"]), seed=1, load_path=str(tmp_path / "ck.pkl"),
                  ops=OracleQuboOps())
    assert np.array_equal(q2.weights_all_visible_to_hidden, q.weights_all_visible_to_hidden)
    assert np.array_equal(q2.weights_hidden_hidden, q.weights_hidden_hidden)


def _check_couplings(q, Q_clamped, Q_unclamped):
    """`_couplings` feeds every training launch: J = off-diagonal of the reference's upper-triangular QUBO, symmetrised,
    in fp32 (src/model/faster_dqbm.py:242-243, :256-271)."""
    for clamped, Qs in ((True, Q_clamped), (False, Q_unclamped)):
        if Qs is None:
            continue
        J = q._couplings(clamped)
        n = Qs.shape[-1]
        assert J.shape == (n, n) and J.dtype == np.float32
        assert np.array_equal(J, J.T) and not np.diag(J).any()
        for Q in Qs:                                   # the quadratic part does not depend on the image
            off = Q - np.diag(np.diag(Q))
            assert not np.tril(off).any()              # the reference's Q is upper triangular
            assert np.array_equal(np.triu(J, 1), off.astype(np.float32))


def test_couplings_match_reference_qubo_offdiagonals():
    g = golden("qbm_small.npz")
    _check_couplings(make(g), g["Q_clamped"], g["Q_unclamped"])            # unrestricted: W_hh and the label row
    g2 = golden("qbm_784.npz")
    from qbm_image_classification_amd.disc_qbm import Disc_QBM

    np.random.seed(99)
    q2 = Disc_QBM(dim_input=784, num_classes=2, n_hidden_nodes=16, seed=int(g2["seed"]), solver="HIP", restricted=True,
                  sample_count=10, anneal_steps=10, beta_eff=1.0, ops=OracleQuboOps())
    _check_couplings(q2, None, g2["Q_unclamped"])                          # restricted 784-input model: label row only
    assert not q2._couplings(True).any()                                   # clamped + restricted: no couplings at all
    # beta_eff scales the couplings like the rest of Q (:245, :283)
    q3 = make(g)
    q3.beta_eff = 2.0
    off = g["Q_unclamped"][0] - np.diag(np.diag(g["Q_unclamped"][0]))
    want = (off + off.T) * float(g["beta_eff"]) / 2.0
    assert np.allclose(q3._couplings(False), want, rtol=1e-7, atol=0)


def test_train_model_dumps_weights_and_history_when_a_step_fails(tmp_path):
    """src/model/faster_dqbm.py:1128-1132: an exception inside the training loop saves the weights and the history before it
    is re-raised -- including the per-batch error / NLL of the minibatches of the CURRENT epoch that did complete (they live
    on the device until the epoch's one synchronisation: ADVICE round 2)."""
    import os

    g = golden("qbm_small.npz")
    q = make(g, ops=OracleQuboOps())
    q.speicherort, q.param_string, q.epochs = str(tmp_path) + "/", "boom", 2
    X, Y = g["X"], g["Y"]
    real, calls = q._step_device, [0]

    def flaky(xb, yb, lr):
        calls[0] += 1
        if calls[0] == 3:
            raise RuntimeError("device lost")
        return real(xb, yb, lr)

    q._step_device = flaky
    with pytest.raises(RuntimeError, match="device lost"):
        q.train_model(X, Y, X, Y, batch_size=1, learning_rate=0.1)
    assert len(q.training_history.nll_per_batch) == 2 and len(q.training_history.errors_per_batch) == 2
    files = os.listdir(str(tmp_path) + "/boom")
    assert any(f.startswith("e1_b3_boom") for f in files), files          # weights at the failing minibatch


def test_assigning_w_hh_or_beta_eff_drops_the_derived_caches():
    g = golden("qbm_small.npz")
    q = make(g, ops=OracleQuboOps())
    q._whh_dev, q._J_cache = "stale", {True: "stale", False: "stale"}
    q.weights_hidden_hidden = q.weights_hidden_hidden * 2.0
    assert q._whh_dev is None and q._J_cache == {}
    q._J_cache = {True: "stale"}
    q.beta_eff = 2.0
    assert q._J_cache == {} and q.beta_eff == 2.0


def test_single_image_statistics_sum_to_the_reference_batch_statistics():
    """get_average_configuration (src/model/faster_dqbm.py:688-752) has no golden of its own; the batched form the golden
    holds (:754-848) is its sum over the images for every block but W_hh (which only the single-image form fills: :739-748,
    checked here against the definition, entry by entry)."""
    g = golden("qbm_small.npz")
    q = make(g, ops=OracleQuboOps({}))
    X, Y = g["X"], g["Y"]
    for tag, S, labels in (("clamped", g["S_clamped"], Y), ("unclamped", g["S_unclamped"], None)):
        tot = None
        for i in range(len(X)):
            samples = [dict(enumerate(map(float, S[i, r]))) for r in range(S.shape[1])]
            out = q.get_average_configuration(samples, X[i], None if labels is None else [labels[i]])
            hid = S[i][:, 1:] if labels is None else S[i]
            for h1 in range(hid.shape[1]):
                for h2 in range(hid.shape[1]):
                    want = np.average(hid[:, h1] * hid[:, h2]) if h2 > h1 else 0.0
                    assert out[5][h1, h2] == pytest.approx(want, abs=1e-15)
            part = [np.asarray(a, dtype=np.float64) for a in out[:5]]
            tot = part if tot is None else [t + a for t, a in zip(tot, part)]
        for nme, arr in zip(["b_h", "b_o", "W_vh", "W_vo", "W_oo"], tot):
            assert np.max(np.abs(arr.reshape(g[f"avg_{tag}_{nme}"].shape) - g[f"avg_{tag}_{nme}"])) <= 1e-12, (tag, nme)


def test_best_epoch_by_combined_accuracy_and_auc():
    g = golden("qbm_small.npz")
    q = make(g, ops=OracleQuboOps({}))
    q.training_history.acc_per_epoch.extend([0.5, 0.8, 0.7])
    q.training_history.auc_per_epoch.extend([0.5, 0.6, 0.9])
    q.training_history.combined_acc_auc_per_epoch.extend([0.5, 0.7, 0.8])
    assert q.get_best_combined_acc_auc() == (2, 0.7, 0.9, 0.8)
