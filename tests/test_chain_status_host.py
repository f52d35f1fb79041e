"""CPU: the host side of the fused chain's fail-safe (rbm_ops.HipRbmOps.check_chain_status) with a faked status word.
No kernel runs here -- the library handle is a stub that only counts calls; what is tested is the bookkeeping: ONE state
buffer that outlives the per-shape workspaces (an epoch with a short last minibatch re-makes the workspace: ADVICE round 2),
which steps are repeated, and that the object leaves the one-launch path."""
import warnings

import pytest
import torch

from qbm_image_classification_amd import rbm_ops
from qbm_image_classification_amd._lib import QBM_U8, QbmHipError


class _StubLib:
    def __init__(self):
        self.mode, self.cd_calls, self.finished = 1, 0, []

    def qbm_rbm_fused_chain_mode(self, m):
        prev = self.mode
        if 0 <= m <= 2:
            self.mode = m
        return prev

    def qbm_rbm_cd_step(self, *a):
        self.cd_calls += 1
        return 0

    def qbm_rbm_cd_step_finish(self, *a):
        self.finished.append((a[8].value if hasattr(a[8], "value") else a[8], a[-2]))   # (draw offset, round)
        return 0


@pytest.fixture()
def ops(monkeypatch):
    monkeypatch.setattr(rbm_ops, "ptr", lambda t: None)                 # CPU tensors stand in for device buffers
    monkeypatch.setattr(rbm_ops, "current_stream", lambda: None)
    o = object.__new__(rbm_ops.HipRbmOps)
    o.lib, o.device = _StubLib(), torch.device("cpu")
    o._disc_ws, o._disc_wss, o._cd_ws, o._cd_wss, o._sync, o._pending, o.status_agree = None, {}, None, {}, None, [], None
    return o


def _P():
    return rbm_ops.ParamSet(torch.zeros(32, 16), torch.zeros(32), torch.zeros(16), torch.zeros(2, 16), torch.zeros(2))


def _step(ops, P, G, B, offset, log):
    v0, y0 = torch.zeros(B, 32, dtype=torch.uint8), torch.zeros(B, 2)

    def run():
        log.append(offset)
        ops.cd_step(P, G, v0, y0, 2, 1, offset, 0, None, None, True, 0.05, 1.0, float(B), 0.0, replay=run)
    run()


def test_status_survives_workspace_rekeying_and_steps_are_replayed(ops):
    P, G, log = _P(), rbm_ops.GradSet(32, 16, 2, "cpu"), []
    for i, B in enumerate((64, 64, 64, 32)):                             # an epoch whose last minibatch is short
        _step(ops, P, G, B, 7 * i, log)
    assert len(ops._cd_wss) == 2
    assert all(w[5] is ops._sync for w in ops._cd_wss.values())          # one state buffer for every cached workspace
    assert [o for o, _, _ in ops._pending] == [0, 7, 14, 21]
    ops._sync[512], ops._sync[513] = 2, 14                               # "the third step's chain gave up (XCD placement)"
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ops.check_chain_status()
    assert log == [0, 7, 14, 21, 14, 21]                                 # the third and the fourth step again, in order
    assert ops.lib.mode == 0 and not ops._sync.any() and ops._pending == []
    assert any("Repeating 2 step" in str(w.message) for w in rec)
    _step(ops, P, G, 64, 28, log)                                        # mode 0: nothing is recorded any more
    assert ops._pending == []


def test_clean_status_clears_the_record_and_checks_come_every_64_steps(ops):
    P, G, log = _P(), rbm_ops.GradSet(32, 16, 2, "cpu"), []
    for i in range(rbm_ops.HipRbmOps.kPendingMax + 3):
        _step(ops, P, G, 64, i, log)
    assert len(ops._pending) == 3                                        # looked at (and cleared) once, at step 64
    ops.check_chain_status()
    assert ops._pending == [] and ops.lib.mode == 1


def test_unknown_failed_step_raises(ops):
    P, G, log = _P(), rbm_ops.GradSet(32, 16, 2, "cpu"), []
    _step(ops, P, G, 64, 5, log)
    ops._sync[512], ops._sync[513] = 1, 999
    with pytest.raises(QbmHipError, match="no longer holds"):
        ops.check_chain_status()


def test_group_agreement_hook_decides(ops):
    """Data-parallel group: this rank's chain was fine, another rank's gave up at offset 3 -> this rank repeats too."""
    P, G, log = _P(), rbm_ops.GradSet(32, 16, 2, "cpu"), []
    for i in range(3):
        _step(ops, P, G, 64, 3 * i, log)
    ops.status_agree = lambda st, off: (1, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ops.check_chain_status()
    assert log == [0, 3, 6, 3, 6] and QBM_U8 == 1


def test_fused_tail_that_gave_up_is_finished_then_the_later_steps_are_repeated(ops):
    """Status 3 (the fused tail of a step -- last hidden phase + gradient + update in one launch -- gave up): part of that
    step's update may have been applied, so it is FINISHED from its workspace (qbm_rbm_cd_step_finish with the recorded
    round), never repeated; the steps behind it, which did nothing, are repeated."""
    P, G, log = _P(), rbm_ops.GradSet(32, 16, 2, "cpu"), []
    for i in range(4):
        _step(ops, P, G, 64, 7 * i, log)
    ops._sync[512], ops._sync[513], ops._sync[515] = 3, 7, 41            # "the second step's fused tail: barrier timed out in round 41"
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ops.check_chain_status()
    assert ops.lib.finished == [(7, 41)]
    assert log == [0, 7, 14, 21, 14, 21]                                 # step 2 finished, steps 3 and 4 again
    assert ops.lib.mode == 0 and not ops._sync.any() and ops._pending == []
    assert any("Finished that step" in str(w.message) and "epeating 2 step" in str(w.message) for w in rec)
