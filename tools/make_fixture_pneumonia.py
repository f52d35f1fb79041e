#!/usr/bin/env python
"""tools/make_fixture_pneumonia.py -- re-encode the reference checkout's PneumoniaMNIST arrays as a committed fixture
(build container only; reads /root/reference/src/data/medmnist/pneumoniamnist.npz, writes tests/golden/data_pneumoniamnist.npz).

Data, not code: the six arrays (train / val / test images uint8 [N,28,28] and labels uint8 [N,1]) are stored unchanged under
the same keys, so that `data_loader.get_medmnist` (reference: src/data_loader.py:115-167) reads the fixture exactly as it reads
the original.  It is the default data set of qbm_main.py (reference: qbm_main.py:36-37, :104) and the data behind the accuracy
context runs of tools/accuracy_context.py; the GPU box receives only /root/repo."""
import os

import numpy as np

SRC = "/root/reference/src/data/medmnist/pneumoniamnist.npz"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "data_pneumoniamnist.npz")

if __name__ == "__main__":
    raw = np.load(SRC)   # allow_pickle=False (the default): plain arrays only
    keys = ("train_images", "train_labels", "val_images", "val_labels", "test_images", "test_labels")
    np.savez_compressed(DST, **{k: raw[k] for k in keys})
    out = np.load(DST)
    for k in keys:
        assert np.array_equal(out[k], raw[k])
        print(k, out[k].shape, out[k].dtype)
    print(os.path.getsize(DST), "bytes")
