"""Multi-GPU: camera streams shard across ranks (one process per GPU), the only exchange is one all-gather of the
per-camera count tensors int32[n_dir, n_cls] over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests).
The reference has no distributed code at all: this is the new merge step of SURVEY.md 8(e).

SURVEY.md 8(f).1, a single stream on several GPUs: the stateless front end (detect + NMS + ReID) shards by frame chunk
(`shard_frames`), the per-detection payloads (box, confidence, class, 512-d feature) are gathered in frame order
(`gather_rows`) and the rank that owns the camera runs the sequential tracker on them."""
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


def ensure_comm(engine):
    """Bring up the engine's RCCL communicator over the ranks of the torch.distributed group (a single process forms a world of
    one): the 128-byte id travels over the group the ranks already share.  Returns (rank, world)."""
    import ctypes as C

    from . import _lib as L
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if not getattr(engine, "_comm_ready", False):
        # RCCL prints a version banner on stdout when it is first used in a process: keep stdout clean for callers that print
        # machine-readable lines (bench.py's one JSON line) by pointing fd 1 at stderr while the communicator comes up
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            idbuf = (C.c_ubyte * 128)()
            if rank == 0:
                L.check(L.lib().vc_comm_unique_id(idbuf))
            if world > 1:
                obj = [bytes(idbuf)]
                dist.broadcast_object_list(obj, src=0)
                idbuf = (C.c_ubyte * 128).from_buffer_copy(obj[0])
            L.check(L.lib().vc_comm_init(engine._h, rank, world, idbuf))
            warm = np.zeros(1, np.int32)                                  # first collective of the communicator (lazy set-up inside RCCL)
            L.check(L.lib().vc_allgather_counts(engine._h, L.ptr(warm, C.c_int), 1, L.ptr(np.zeros(world, np.int32), C.c_int)))
        finally:
            C.CDLL(None).fflush(None)                                     # the banner sits in libc's stdout buffer: flush it while fd 1 is stderr
            os.dup2(saved, 1)
            os.close(saved)
        engine._comm_ready = True
    return rank, world


def allgather_counts_native(engine, local_counts):
    """The count all-gather through the C ABI (vc_comm_init / vc_allgather_counts: ncclAllGather on the engine's stream, RCCL over
    xGMI).  Returns int32 (world * n_cam_local, n_dir, n_cls), rank-major."""
    import ctypes as C

    from . import _lib as L
    t = np.ascontiguousarray(local_counts, dtype=np.int32)
    rank, world = ensure_comm(engine)
    out = np.zeros((world,) + t.shape, np.int32)
    L.check(L.lib().vc_allgather_counts(engine._h, L.ptr(t.reshape(-1), C.c_int), t.size, L.ptr(out.reshape(-1), C.c_int)))
    return out.reshape((world * t.shape[0],) + t.shape[1:])


def share_tune_cache(engine, warm=None, src=0):
    """Rank `src` runs `warm()` (its conv autotune: one pass over every shape the run will use), exports the choices and broadcasts
    them; every other rank imports them BEFORE its own first launches.  N ranks then time the candidates once instead of N times, and
    -- what matters for parity -- all ranks run the same tile configuration for every layer (two members of a near-tie from different
    kernel families may differ in the last bf16 bit).  Single process: just runs `warm`.  Returns the text."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if rank == src and warm is not None:
        warm()
    text = engine.tune_export() if rank == src else None
    if world > 1:
        obj = [text]
        dist.broadcast_object_list(obj, src=src)
        text = obj[0]
        if rank != src:
            engine.tune_import(text)
    return text


def gather_compact(padded, counts):
    """vc_gather_compact_host through ctypes: padded (world, max_rows, ...) rank-major blocks -> the first sum(counts) rows."""
    import ctypes as C

    from . import _lib as L
    a = np.ascontiguousarray(padded)
    world, max_rows = a.shape[0], a.shape[1]
    row_bytes = int(a.dtype.itemsize * int(np.prod(a.shape[2:], dtype=np.int64)))
    cnt = np.ascontiguousarray(counts, dtype=np.int32)
    out = np.zeros((world * max_rows,) + a.shape[2:], a.dtype)
    total = C.c_int64()
    L.check(L.lib().vc_gather_compact_host(C.c_void_p(a.ctypes.data), world, max_rows, row_bytes, L.ptr(cnt, C.c_int), C.c_void_p(out.ctypes.data),
                                           world * max_rows, C.byref(total)))
    return out[: total.value]


def shard_streams(n_streams, rank, world):
    """Stream i is owned by rank i % world (whole streams only: tracker state never shards below a camera)."""
    return [i for i in range(n_streams) if i % world == rank]


def shard_frames(n_frames, rank, world, chunk):
    """Chunk j = frames [j*chunk, (j+1)*chunk) belongs to rank j % world.  Returns the (start, stop) chunks of `rank`, in order;
    round r of every rank covers the consecutive chunks r*world .. r*world + world - 1."""
    out = []
    for j, s in enumerate(range(0, n_frames, chunk)):
        if j % world == rank:
            out.append((s, min(s + chunk, n_frames)))
    return out


def gather_rows(local_rows, device=None):
    """local_rows: float64 (n, C) whose column 0 is a frame index.  Every rank receives the rows of all ranks ordered by frame
    index (rows of one frame keep their rank-local order; a frame lives on exactly one rank).  Two collectives: the row counts,
    then the rows padded to the largest count (RCCL all_gather wants equal shapes)."""
    rows = np.ascontiguousarray(local_rows, dtype=np.float64)
    if rows.ndim != 2:
        raise ValueError("gather_rows expects a 2-D array")
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rows[np.argsort(rows[:, 0], kind="stable")] if len(rows) else rows
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64)
    if device is not None:
        n = n.to(device)
    counts = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = torch.zeros((cap, rows.shape[1]), dtype=torch.float64)
    pad[: rows.shape[0]] = torch.from_numpy(rows)
    if device is not None:
        pad = pad.to(device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    allr = np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], 0)
    return allr[np.argsort(allr[:, 0], kind="stable")] if len(allr) else allr
