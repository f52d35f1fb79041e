"""GPU: the two drivers north_star tells the build to keep -- rbm_main.py (reference: rbm_main.py:36-82) and qbm_main.py
(reference: qbm_main.py:23-110, :138-218) -- EXECUTED in-process through their own argument parsers: flag parsing, the
DeviceDataset / --host-loader selection, the rank-0 pickles, --bars-and-stripes (BASELINE.json configs[0] as worded:
"via rbm_main.py") and the result writing of qbm_main."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from helpers import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(tmp, seed):
    with open(os.path.join(tmp, f"acc_per_epoch{seed}.pkl"), "rb") as f:
        acc = pickle.load(f)
    with open(os.path.join(tmp, f"auc_per_epoch{seed}.pkl"), "rb") as f:
        auc = pickle.load(f)
    return acc, auc


def test_rbm_main_bars_and_stripes_config1(tmp_path):
    """BASELINE.json configs[0]: Bars-and-Stripes 4x4 -> ClassificationRBM 16v / 8h, CD-1, via rbm_main.py."""
    import rbm_main

    args = rbm_main.parser.parse_args(["--bars-and-stripes", "--type", "generative", "--cd-k", "1", "--epochs", "3", "--seeds", "1",
                                       "--hidden-units", "8", "--out-dir", str(tmp_path)])
    res = rbm_main.main(args)
    assert args.visible_units == 16 and len(res) == 1
    seed = res[0][0]
    assert seed == 1967690937                                    # the first seed of rbm_main.py:32-33
    acc, auc = _load(str(tmp_path), seed)
    assert len(acc) == 3 and len(auc) == 3 and all(0.0 <= a <= 1.0 for a in acc)
    # the hybrid rule LEARNS the task through the driver (generative alone is not asked to classify)
    args = rbm_main.parser.parse_args(["--bars-and-stripes", "--type", "hybrid", "--cd-k", "1", "--epochs", "150", "--seeds", "1",
                                       "--hidden-units", "8", "--lr", "0.5", "--generative-factor", "0.1", "--out-dir", str(tmp_path)])
    res = rbm_main.main(args)
    assert res[0][1] >= 0.95, res


@pytest.mark.parametrize("host_loader", [True, False])
def test_rbm_main_default_flow_breastmnist(tmp_path, host_loader):
    """The default command line of the reference (discriminative rule, H = 9, batch 7, BreastMNIST) for 1 seed x 2 epochs.
    --host-loader is the reference's DataLoader(shuffle=True): accuracy and AUC of the first two epochs equal the imported
    reference's run (tools/make_golden_runs.py).  The default loader is the device-resident dataset (another permutation):
    the pickles exist and hold sane values."""
    import rbm_main

    argv = ["--epochs", "2", "--seeds", "1", "--out-dir", str(tmp_path)] + (["--host-loader"] if host_loader else [])
    res = rbm_main.main(rbm_main.parser.parse_args(argv))
    seed = res[0][0]
    acc, auc = _load(str(tmp_path), seed)
    assert len(acc) == 2 and len(auc) == 2
    if host_loader:
        g = golden("rbm_run_breast_h9.npz")
        assert np.array_equal(acc, g[f"s{seed}_acc"][:2]), (acc, g[f"s{seed}_acc"][:2])
        assert np.array_equal(auc, g[f"s{seed}_auc"][:2])
    else:
        assert all(0.5 <= a <= 1.0 for a in acc)                  # majority class of the test split: 0.73
    assert res[0][1] == acc[-1] and res[0][2] == auc[-1]


@pytest.mark.parametrize("data_set", ["breastmnist", "mnist"])
def test_qbm_main_function(tmp_path, data_set):
    """qbm_main.main(...) (reference qbm_main.py:23-110): data set -> Disc_QBM(solver HIP) -> train_model -> predict ->
    metrics; the result pickle and the history are written."""
    import qbm_main

    save = str(tmp_path) + "/"
    acc, auc = qbm_main.main(seed=7, n_hidden_nodes=4, solver="HIP", sample_count=20, anneal=20, epochs=1, batch_size=16,
                             learning_rate=0.1, restricted=False, data_set=data_set, save=save, name="t_", max_train=64)
    assert 0.0 <= acc <= 1.0 and 0.0 <= auc <= 1.0
    with open(save + "t_acc_auc.pkl", "rb") as f:
        assert pickle.load(f) == (acc, auc)
    assert any(n.startswith("t_") and n != "t_acc_auc.pkl" for n in os.listdir(save))   # metrics.save_history wrote the history


def test_qbm_main_command_line_defaults(tmp_path):
    """The script as a process with its DEFAULT data set (pneumoniamnist: the committed fixture
    tests/golden/data_pneumoniamnist.npz) -- only the sizes are cut down so that it finishes in seconds."""
    out = str(tmp_path) + "/"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "qbm_main.py"), "-e", "1", "-sc", "10", "--anneal", "10", "--max_train", "73",
                        "--load_path", out, "--name", "cli_"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Accuracy:" in r.stdout and os.path.exists(out + "cli_acc_auc.pkl")
