"""GPU: the data-parallel PRODUCT path on HIP + RCCL.

What is sharded: the sum over the rows of a minibatch in the gradient (reference: src/ClassificationRBM.py:119-136;
src/model/faster_dqbm.py:1042-1049 for the QBM).  This pool hands the build ONE GPU, so the nearest thing to N > 1 is a
one-rank `nccl` group: the very host code of N ranks (cd_step without the fused update -> RCCL all-reduce of the flat
gradient buffer, row count included -> qbm_rbm_update_dev reading the count on the device), real RCCL, and a result
that must equal the single-GPU object's bit for bit because a one-rank all-reduce adds nothing.  The group has to exist
before the process touches the GPU, hence a child process (tests/dp_nccl_child.py); the gloo tests of
tests/test_data_parallel.py cover world sizes 2 and 4 on the CPU.  N > 1 on hardware stays unmeasured."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def test_product_path_under_a_one_rank_nccl_group_equals_single_gpu(tmp_path):
    """ClassificationRBM(data_parallel=True).train_rbm -- discriminative, CD-1, PCD-2; 3 epochs, ragged last minibatch;
    epochs captured in one hipGraph (RCCL inside) and eager -- and Disc_QBM(data_parallel=True).train_model: parameters,
    persistent chains, loss lists and draw counters bit-equal to the objects without the group."""
    out = tmp_path / "dp.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=os.pathsep.join([os.path.dirname(HERE), HERE, os.environ.get("PYTHONPATH", "")]))
    env.pop("QBM_EPOCH_GRAPH", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "dp_nccl_child.py"), str(out)], env=env, capture_output=True, text=True, timeout=400)
    assert out.exists(), f"child died (rc {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    res = json.loads(out.read_text())
    assert res["ok"], res.get("error", "") + "\n" + res.get("trace", "") + "\n" + r.stderr[-2000:]
    assert r.returncode == 0
    assert res["backend"] == "nccl" and res["world"] == 1
    assert len(res["compared"]) == 7, res["compared"]      # 3 rules x (graph, eager) + Disc_QBM
