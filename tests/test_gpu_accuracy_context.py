"""GPU: CONTEXT, PARITY UNPINNED.  The reference's published PneumoniaMNIST rows (src/paper_plots/test_result.txt:1-15,
out/paper_data/QuCUN_Pneumonia_q/third_pneu_result.txt:1-15, out/BA_data/RBM_Pneumonia_results/
rbm_performance_summary_Pneumonia.csv:2,6) at their hyper-parameters with this build's sampler / trainer
(tools/accuracy_context.py; the same three seeds per row as profiles/r04_accuracy_context.jsonl).  The reference's annealer
(dwave-neal / a QPU) is absent, so its sample values cannot be reproduced: the bar here is the accuracy REGIME -- the MEAN test
accuracy of three seeds must land within 0.03 of the published row (round 3 asked one seed for "published - 0.05 or above"),
and every run well above the majority-class rate of the test split (0.625).  One row is held from below only: the
ClassificationRBM with 20 hidden units, whose published mean of ten seeds (0.7877) this build's three seeds exceed by 0.05
(0.795 / 0.857 / 0.862: the reference's own 14-hidden row, 0.8354, is where they sit)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


SEEDS = (19, 23, 29)


def _check(which, runs, two_sided=True):
    import numpy as np

    import accuracy_context as ctx

    ref_acc, _, _ = ctx.REFERENCE[which]
    mean = float(np.mean([r["test_acc"] for r in runs]))
    assert mean >= ref_acc - 0.03, (mean, runs)            # context bar, not parity
    if two_sided:
        assert mean <= ref_acc + 0.03, (mean, runs)
    for r in runs:
        assert r["test_acc"] >= 0.625 + 0.15 and r["test_auc"] >= 0.70, r


@pytest.mark.parametrize("which,beta1", [("qbm_h2", False), ("qbm_h2", True), ("qbm_h10", False)])
def test_disc_qbm_pneumonia_rows(which, beta1):
    import accuracy_context as ctx

    _check(which, [ctx.run_qbm(which, seed=s, beta1=beta1) for s in SEEDS])


@pytest.mark.parametrize("which", ["rbm_h14", "rbm_h20"])
def test_classification_rbm_pneumonia_rows(which):
    import accuracy_context as ctx

    _check(which, [ctx.run_rbm(which, seed=s) for s in ctx.RBM_SEEDS[:3]], two_sided=which != "rbm_h20")
