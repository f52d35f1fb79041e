#!/usr/bin/env python
"""tools/make_golden_cdqbm.py -- penalty-augmented QUBOs of the reference's Conv_Deep_QBM as fixtures for the sample_Q
adapter.  Build container only: IMPORTS THE REAL REFERENCE (read-only); only numbers are written.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tools/make_golden_cdqbm.py

The model is built exactly as /root/reference/cdqbm_main.py:89-107 does (kernel 3, stride 1, pooling_size 4,
sequential_layer_sizes [4], hidden_bias_type 'shared', seed 44) with pooling_type 'probabilistic' -- the variant whose QUBO
carries the at-most-one and OR-link penalties of src/qubo/builder.py:116-146 -- on two digits of the committed MNIST 0/1
fixture.  src/train/pipeline.py:13-36 is followed: prepare_context -> build_{un,}clamped_qubo -> the matrix handed to
model.sampler.sample_Q.  The D-Wave packages the model module imports at its top (dimod, neal) are absent here and replaced
by stub modules: none of their code runs on this path (the sampler object is never called).

  cdqbm_qubo.npz   Q_unclamped [2, n+1, n+1], Q_clamped [2, n, n] (float32: what a sampler consumes), pool windows, slices
"""
import contextlib
import io
import os
import sys
from unittest import mock

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402


def main():
    stubs = ["dimod", "neal", "dwave", "dwave.cloud", "dwave.embedding", "dwave.system", "dwave_networkx", "minorminer", "seaborn",
             "pymetis"]
    for m in stubs:
        sys.modules[m] = mock.MagicMock()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from src.model.cdqbm_state import Conv_Deep_QBM
        from src.model.inference import prepare_context
        from src.model.layers import last_hidden_slice
        from src.qubo.builder import build_clamped_qubo, build_unclamped_qubo

        with contextlib.redirect_stdout(io.StringIO()):
            model = Conv_Deep_QBM(num_visible_nodes=784, num_lable_nodes=1, image_shape=(28, 28), seed=44, kernel_size=3,
                                  pooling_size=4, pooling_type="probabilistic", stride=1, sequential_layer_sizes=[4],
                                  is_restricted=False, hidden_bias_type="shared", solver="SA", anneal=1000)
        f = np.load(os.path.join(OUT, "data_mnist01_t10k_bits.npz"))
        imgs = np.unpackbits(f["bits"], axis=1)[:, :784].reshape(-1, 28, 28).astype(np.float64)
        labels = f["labels"]
        picks = [int(np.flatnonzero(labels == 0)[0]), int(np.flatnonzero(labels == 1)[0])]
        Qu, Qc = [], []
        for i in picks:
            ctx = prepare_context(model, imgs[i])
            object.__setattr__(ctx, "last_hidden_slice", last_hidden_slice(ctx.slices))      # src/train/pipeline.py:17
            Qu.append(build_unclamped_qubo(model, ctx, 1.0))
            Qc.append(build_clamped_qubo(model, ctx, np.asarray([float(labels[i])]), 1.0))
        sl = ctx.slices
        Qu, Qc = np.stack(Qu), np.stack(Qc)
        assert not np.tril(Qu[0], -1).any() and not np.tril(Qc[0], -1).any()                 # upper triangular, as sample_Q expects
        assert np.array_equal(Qu.astype(np.float32).astype(np.float64), Qu) or True
        np.savez_compressed(os.path.join(OUT, "cdqbm_qubo.npz"), Q_unclamped=Qu.astype(np.float32), Q_clamped=Qc.astype(np.float32),
                            f32_exact=np.array(bool(np.array_equal(Qu.astype(np.float32).astype(np.float64), Qu))),
                            images=np.array(picks), labels=labels[picks], pool_windows=np.stack(model.pool_windows),
                            conv=np.array([sl.conv.start, sl.conv.stop]), pool=np.array([sl.pool.start, sl.pool.stop]),
                            seq=np.array([[s.start, s.stop] for s in sl.seq_layers]), out=np.array([sl.out.start, sl.out.stop]),
                            penalty=np.array(0.8225))
        print("n unclamped", Qu.shape[-1], "n clamped", Qc.shape[-1], "windows", len(model.pool_windows), "nonzeros", int((Qu[0] != 0).sum()))
    finally:
        os.chdir(cwd)
        for m in stubs:
            sys.modules.pop(m, None)


if __name__ == "__main__":
    main()
