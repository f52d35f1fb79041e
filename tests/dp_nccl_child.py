"""Child process of tests/test_gpu_data_parallel.py: the data-parallel PRODUCT path on HIP + RCCL.

    python tests/dp_nccl_child.py OUT.json

A ONE-rank `nccl` (= RCCL) process group is created before anything else touches the GPU (the nearest thing to N > 1 a
one-GPU box allows: the same host code, the same launches, a real RCCL all-reduce on the flat gradient buffer -- over one
rank it adds nothing, so every result must equal the single-GPU object's BIT FOR BIT).  Scenarios:
  * ClassificationRBM(data_parallel=True).train_rbm -- discriminative (src/ClassificationRBM.py:101-146 under the loop
    :159-205), CD-1 and PCD-2 -- 3 epochs from a DeviceDataset with a ragged last minibatch, with the
    epoch captured in one hipGraph (QBM_EPOCH_GRAPH=1: RCCL all-reduce + device-count update inside the graph) and eagerly
    (=0): parameters, persistent chains, loss / NLL lists, draw counters;
  * Disc_QBM(data_parallel=True).train_model (src/model/faster_dqbm.py:1079-1166; the sum over images it shards:
    :1042-1049), 2 epochs: theta and the history.
Writes {"ok": true, "compared": [...]} or the first mismatch to OUT.json; exit code 0 only when everything matched."""
import contextlib
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out_path):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("QBM_DP_TEST_PORT", "29547"), RANK="0", WORLD_SIZE="1")
    import numpy as np
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)   # (binds the device; the group is created before any kernel, copy or allocation of ours)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    compared = []
    try:
        from qbm_image_classification_amd.classification_rbm import ClassificationRBM
        from qbm_image_classification_amd.device_data import DeviceDataset
        from qbm_image_classification_amd.disc_qbm import Disc_QBM

        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        rs = np.random.RandomState(9)
        V, H, C, N, B = 784, 64, 3, 150, 32            # 5 minibatches per epoch, the last one of 22 rows
        x = (rs.random_sample((N, V)) < 0.25).astype(np.float32)
        y = rs.randint(0, C, N)

        def run(method, k, pcd, dp, flag):
            os.environ["QBM_EPOCH_GRAPH"] = flag
            m = ClassificationRBM(V, H, k, num_classes=C, learning_rate=0.05, seed=2, device="cuda:0", persistent=pcd,
                                  data_parallel=dp)
            assert m.data_parallel == dp
            ds = DeviceDataset(x, y, batch_size=B, num_classes=C, storage="f32" if method == "discriminative" else "bits", seed=4)
            with contextlib.redirect_stdout(io.StringIO()):
                loss, _, nll = m.train_rbm(ds, epochs=3, method=method)
            torch.cuda.synchronize()
            out = dict(loss=loss, nll=nll, draw=m._draw_offset, rows=m._rows_seen, graph=m._epoch_graph is not None)
            for nm in ("weights", "visible_bias", "hidden_bias", "class_weights", "class_bias"):
                out[nm] = getattr(m, nm).clone()
            if pcd:
                out["chain_v"], out["chain_y"] = m._chain_v.clone(), m._chain_y.clone()
            return out

        for method, k, pcd in (("discriminative", 1, False), ("generative", 1, False), ("generative", 2, True)):
            ref = run(method, k, pcd, False, "0")      # the single-GPU object, eager epochs
            assert len(set(ref["loss"])) == 3, "the epochs must differ"
            for flag in ("1", "0"):
                got = run(method, k, pcd, True, flag)
                name = f"{method}{' PCD-2' if pcd else (' CD-1' if method == 'generative' else '')} data_parallel QBM_EPOCH_GRAPH={flag}"
                # k >= 2 takes the one-launch chain, whose fail-safe record is per step on the host: those epochs stay eager
                want_graph = flag == "1" and not pcd
                if got["graph"] != want_graph:
                    raise AssertionError(f"{name}: epoch graph captured = {got['graph']}, expected {want_graph}")
                for key, a in ref.items():
                    b = got[key]
                    if key == "graph":
                        continue
                    same = torch.equal(a, b) if torch.is_tensor(a) else a == b
                    if not same:
                        raise AssertionError(f"{name}: `{key}` differs from the single-GPU object")
                compared.append(name)

        # ---- Disc_QBM under the group: every rank is handed the same minibatch, samples its rows, ONE float64 all-reduce
        Dq, Hq = 64, 8
        Xq = (rs.random_sample((45, Dq)) < 0.3).astype(np.float64)
        Yq = rs.randint(0, 2, 45)
        res = []
        for dp in (False, True):
            np.random.seed(1)
            q = Disc_QBM(dim_input=Dq, num_classes=2, n_hidden_nodes=Hq, seed=3, solver="HIP", sample_count=12, anneal_steps=6,
                         epochs=2, speicherort=os.path.join(os.path.dirname(out_path), f"qbm_dp{int(dp)}_"), param_string="t",
                         data_parallel=dp)
            with contextlib.redirect_stdout(io.StringIO()):
                q.train_model(Xq, Yq, Xq[:20], Yq[:20], batch_size=10, learning_rate=0.05)   # 5 minibatches, the last of 5 images
            torch.cuda.synchronize()
            h = q.training_history
            res.append((q._theta.clone(), list(h.nll_per_batch), list(h.errors_per_batch), list(h.acc_per_epoch)))
        if not torch.equal(res[0][0], res[1][0]):
            raise AssertionError("Disc_QBM.train_model: theta under the 1-rank group differs from the single-GPU object "
                                 f"(max |diff| {float((res[0][0] - res[1][0]).abs().max()):.3e})")
        if res[0][1:] != res[1][1:]:
            raise AssertionError("Disc_QBM.train_model: training history differs under the 1-rank group")
        compared.append("Disc_QBM.train_model data_parallel (theta bit-equal: a one-rank all-reduce reorders nothing)")
        result = {"ok": True, "compared": compared, "backend": dist.get_backend(), "world": dist.get_world_size()}
    except Exception as e:   # noqa: BLE001
        import traceback

        result = {"ok": False, "compared": compared, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-3000:]}
    finally:
        dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    return 0 if result["ok"] else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
