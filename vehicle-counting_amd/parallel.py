"""Multi-GPU: camera streams shard across ranks (one process per GPU), the only exchange is one all-gather of the
per-camera count tensors int32[n_dir, n_cls] over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).
The reference has no distributed code at all: this is the new merge step of SURVEY.md 8(e)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Rendezvous from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torchrun sets them.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def counts_to_tensor(counts, direction_keys, num_classes):
    t = np.zeros((len(direction_keys), num_classes), np.int32)
    for i, d in enumerate(direction_keys):
        t[i] = counts[d]
    return t


def allgather_counts(local_counts, device=None):
    """local_counts: int32 (n_cam_local, n_dir, n_cls) -> (world * n_cam_local, n_dir, n_cls), rank-major."""
    t = torch.as_tensor(np.ascontiguousarray(local_counts, dtype=np.int32))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t.numpy().copy()
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def shard_streams(n_streams, rank, world):
    """Stream i is owned by rank i % world (whole streams only: tracker state never shards below a camera)."""
    return [i for i in range(n_streams) if i % world == rank]
