"""Coreset selection: the HIP mirror of the reference's `src/coreset.py`.

`k_center_greedy` keeps the reference's signature and bookkeeping (`src/coreset.py:28-106`: `budget += 1`, the
random first centre from `np.random.seed(random_seed)`, `index[select_mask]` as the result) and runs the picks on
the GPU through `qbm_kcenter_greedy_fused` (all picks in one launch: 256 persistent workgroups with their rows in LDS) or,
for shapes that kernel does not take, `qbm_kcenter_greedy` (one launch per pick).  Only the Euclidean metric of the reference
(`euclidean_dist`, `:6-13`) is implemented; a custom `metric` callable is refused rather than silently ignored.
`random_downsampling` (`:123-141`) is host-side numpy, as in the reference.  No CPU fallback for the picks."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr


def euclidean_dist(x, y):
    """Pairwise Euclidean distances [m, n] between the rows of x [m, d] and y [n, d], floored at 1e-6 (= sqrt(1e-12)),
    for callers that import the reference's helper of this name (src/coreset.py:6-13); torch, any device.  The picks
    themselves never call it: they run in csrc/coreset_kernels.hip."""
    sq = (x * x).sum(dim=1)[:, None] + (y * y).sum(dim=1)[None, :] - 2.0 * torch.mm(x, y.transpose(0, 1))
    return torch.sqrt(torch.clamp(sq, min=1e-12))


def kcenter_picks(emb, seeds, n_picks, device="cuda:0", fused=None):
    """Device picks for given seed centres -> (picks int32 [n_picks], selected uint8 [N]) as device tensors.
    fused: True / False = the one-launch kernel (qbm_kcenter_greedy_fused) / one launch per pick; None = the one-launch kernel
    where the shape fits it, and -- should that launch report that its workgroups were not all resident -- the selection is
    repeated with one launch per pick (same contract, same picks)."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.QbmHipError("no ROCm device visible: the HIP path cannot run (there is no CPU fallback)")
    dev = torch.device(device)
    emb = torch.as_tensor(emb).to(dev, torch.float32).contiguous()
    n, d = emb.shape
    seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    if fused is None or fused:
        seeds_dev = torch.from_numpy(seeds).to(dev)
        ws = torch.empty(lib.qbm_kcenter_fused_workspace_bytes(), dtype=torch.uint8, device=dev)
        picks = torch.empty(max(n_picks, 1), dtype=torch.int32, device=dev)
        selected = torch.empty(n, dtype=torch.uint8, device=dev)
        rc = lib.qbm_kcenter_greedy_fused(ptr(emb), n, d, ptr(seeds_dev), len(seeds), n_picks, ptr(picks), ptr(selected), ptr(ws),
                                          current_stream())
        if rc == 0:
            status = int(ws.view(torch.int32)[9].item())      # (synchronises: the caller reads the picks next anyway)
            if status == 0:
                return picks[:n_picks], selected
            if fused:
                raise _lib.QbmHipError(f"one-launch k-center kernel gave up (status {status}): its workgroups were not all resident")
            import warnings

            warnings.warn(f"one-launch k-center kernel gave up (status {status}); repeating the selection with one launch per pick",
                          RuntimeWarning)
        elif fused:
            check(rc, "qbm_kcenter_greedy_fused")
    embT = emb.t().contiguous()
    ws = torch.empty(lib.qbm_kcenter_workspace_bytes(n, len(seeds)), dtype=torch.uint8, device=dev)
    picks = torch.empty(max(n_picks, 1), dtype=torch.int32, device=dev)
    selected = torch.empty(n, dtype=torch.uint8, device=dev)
    check(lib.qbm_kcenter_greedy(ptr(embT), n, d, seeds.ctypes.data_as(C.POINTER(C.c_int32)), len(seeds), n_picks, ptr(picks),
                                 ptr(selected), ptr(ws), current_stream()), "qbm_kcenter_greedy")
    return picks[:n_picks], selected


def k_center_greedy(embedding_matrix, budget: int, metric=euclidean_dist, device=None, random_seed=None, index=None,
                    already_selected=None, print_freq: int = 20):
    """src/coreset.py:28-106 -> index[select_mask] (numpy), exactly `budget` entries when budget < N."""
    if isinstance(embedding_matrix, np.ndarray):
        assert embedding_matrix.ndim == 2
    elif isinstance(embedding_matrix, torch.Tensor):
        assert embedding_matrix.dim() == 2
    else:
        raise TypeError("embedding_matrix must be np.ndarray or torch.Tensor")
    if not callable(metric):
        raise ValueError("metric must be callable")
    if metric is not euclidean_dist:
        raise NotImplementedError("the HIP k-center path implements the reference's euclidean_dist only")
    budget += 1
    sample_num = int(embedding_matrix.shape[0])
    if budget < 0:
        raise ValueError("Illegal budget size.")
    if budget > sample_num:
        budget = sample_num
    if index is not None:
        assert sample_num == len(index)
        index = np.asarray(index)
    else:
        index = np.arange(sample_num)
    already_selected = [] if already_selected is None else list(already_selected)
    np.random.seed(random_seed)
    select_mask = np.zeros(sample_num, dtype=bool)
    if len(already_selected) == 0:
        already_selected = [np.random.randint(0, sample_num)]
    select_mask[already_selected] = True
    remaining_budget = budget - int(select_mask.sum())
    if remaining_budget <= 0:
        return index[select_mask]
    dev = "cuda:0" if device is None or torch.device(device).type != "cuda" else device
    _, selected = kcenter_picks(embedding_matrix, np.flatnonzero(select_mask), remaining_budget, dev)
    return index[selected.cpu().numpy().astype(bool)]


def downsample_kcenter(embeddings, train_x, train_y, budget: int, random_seed: int):
    """src/coreset.py:109-120 with the embeddings passed in (the reference takes them from Conv_Deep_QBM, out of scope;
    for the RBM / Disc_QBM path they are the hidden means p(h|x))."""
    idx = k_center_greedy(embeddings, budget=budget, random_seed=random_seed)
    return train_x[idx], train_y[idx]


def random_downsampling(x, y, downsample_size, seed: int = 42):
    """src/coreset.py:123-141"""
    if len(x) < downsample_size:
        raise ValueError("Downsample size must be less than the number of samples in the dataset.")
    np.random.seed(seed)
    indices = np.random.choice(len(x), downsample_size, replace=False)
    return x[indices], y[indices]
