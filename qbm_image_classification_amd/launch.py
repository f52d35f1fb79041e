"""One process per GPU: `python -m torch.distributed.run --nproc-per-node N rbm_main.py ...` (or qbm_main.py).

`init_distributed()` reads RANK / LOCAL_RANK / WORLD_SIZE, binds the process to its GPU and creates the RCCL process
group (`backend="nccl"` is RCCL on ROCm).  With WORLD_SIZE unset or 1 it does nothing.  `ClassificationRBM` and
`Disc_QBM` detect the initialised group and shard every minibatch over its ranks."""
from __future__ import annotations

import os


def init_distributed():
    """-> (rank, world_size, device string)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, "cuda:0"
    import torch
    import torch.distributed as dist

    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    return rank, world, f"cuda:{local}"


def finalize_distributed():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
