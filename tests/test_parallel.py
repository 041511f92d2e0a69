"""CPU: the N>1 path -- stream sharding and the single count all-gather -- with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from vehicle_counting_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_from_env("gloo")
    streams = parallel.shard_streams(5, r, w)
    # per-camera counts int32[n_dir=2, n_cls=3], value encodes (stream, dir, cls)
    local = np.stack([np.arange(6, dtype=np.int32).reshape(2, 3) + 100 * s for s in streams[:2]])
    out = parallel.allgather_counts(local)
    q.put((r, streams, out))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_counts_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    base = np.arange(6, dtype=np.int32).reshape(2, 3)
    expect = np.stack([base + 0, base + 200, base + 100, base + 300])        # rank-major: rank0 streams 0,2; rank1 streams 1,3
    for _, _, out in res:
        np.testing.assert_array_equal(out, expect)


def test_single_process_passthrough():
    x = np.arange(12, dtype=np.int32).reshape(1, 4, 3)
    np.testing.assert_array_equal(parallel.allgather_counts(x), x)
    assert parallel.shard_streams(8, 3, 8) == [3]
    c = parallel.counts_to_tensor({"01": [1, 2], "02": [3, 4]}, ["01", "02"], 2)
    np.testing.assert_array_equal(c, [[1, 2], [3, 4]])


def _worker_frames(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = parallel.init_from_env("gloo")
    mine = parallel.shard_frames(23, r, w, chunk=4)
    rounds = []
    for k in range(3):                                            # 6 chunks over 2 ranks = 3 rounds
        local = []
        if k < len(mine):
            for f in range(*mine[k]):
                n = (f * 7) % 4                                   # 0..3 detections per frame, some frames empty
                for d in range(n):
                    local.append([f + 1, 1000 * r + 10 * f + d] + [0.5] * 3)
        rows = parallel.gather_rows(np.asarray(local, np.float64).reshape(-1, 5))
        rounds.append(rows)
    q.put((r, mine, rounds))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_gather_two_ranks():
    """frame chunks alternate over the ranks; the gathered detection rows arrive on every rank ordered by
    frame id with the rank-local order inside a frame, uneven (and empty) contributions included."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_frames, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [(0, 4), (8, 12), (16, 20)] and res[1][1] == [(4, 8), (12, 16), (20, 23)]
    for k in range(3):
        a, b = res[0][2][k], res[1][2][k]
        np.testing.assert_array_equal(a, b)                        # every rank holds the same gathered rows
        frames = a[:, 0]
        assert np.all(np.diff(frames) >= 0)
        lo = 8 * k + 1
        assert frames.min() >= lo and frames.max() <= min(lo + 7, 23)
        for f in np.unique(frames):                                # inside a frame: the owner's local order
            d = a[frames == f][:, 1]
            owner = ((int(f) - 1) // 4) % 2
            np.testing.assert_array_equal(d, [1000 * owner + 10 * (int(f) - 1) + i for i in range(len(d))])
        expect = sum(((f * 7) % 4) for f in range(8 * k, min(8 * k + 8, 23)))
        assert len(a) == expect
    # single process: identity up to the stable sort
    x = np.array([[3, 1.0], [1, 2.0], [3, 3.0], [2, 4.0]])
    np.testing.assert_array_equal(parallel.gather_rows(x)[:, 1], [2.0, 4.0, 1.0, 3.0])


# ---- vc_allgather_rows' host half for worlds no single-GPU box can run -----------------------------------------
import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gather_compact_layout_for_worlds_2_3_8(world):
    """The pad / compact logic of vc_allgather_rows (RCCL gathers equal blocks: every rank pads to the largest contribution) on the
    buffers ranks of a world of 2 / 3 / 8 would produce, uneven and EMPTY contributions included: the compacted rows are the ranks'
    rows in rank-major order = frame order for chunks dealt round-robin (parallel.shard_frames), for the float64 rows and for the
    float32 [512] embeddings alike (same offsets: vc_gather_offsets drives the device-side copies)."""
    import ctypes as C

    from vehicle_counting_amd import _lib as L
    rng = np.random.default_rng(world)
    for trial in range(20):
        counts = rng.integers(0, 9, world)
        counts[rng.integers(0, world)] = 0                                     # at least one rank has nothing this round
        if trial == 0:
            counts[:] = 0                                                      # nobody has anything
        if trial == 1:
            counts[:] = 5                                                      # no padding at all
        maxn = int(max(counts.max(), 1)) if trial % 2 else int(counts.max())   # the C side sizes blocks by the true maximum
        rows = [np.column_stack([np.full(c, 100 * r + 1.0), rng.standard_normal((c, 6))]) if c else np.zeros((0, 7)) for r, c in enumerate(counts)]
        feats = [rng.standard_normal((c, 512)).astype(np.float32) for c in counts]
        pad_rows = np.full((world, maxn, 7), -7.0)                             # padding is garbage that must never surface
        pad_feat = np.full((world, maxn, 512), np.float32(-9.0))
        for r, c in enumerate(counts):
            pad_rows[r, :c] = rows[r]
            pad_feat[r, :c] = feats[r]
        got_rows = parallel.gather_compact(pad_rows, counts)
        got_feat = parallel.gather_compact(pad_feat, counts)
        np.testing.assert_array_equal(got_rows, np.concatenate(rows, 0))
        np.testing.assert_array_equal(got_feat, np.concatenate(feats, 0))
        src, dst, tot = np.zeros(world, np.int64), np.zeros(world, np.int64), C.c_int64()
        cnt = np.ascontiguousarray(counts, np.int32)
        L.check(L.lib().vc_gather_offsets(world, maxn, L.ptr(cnt, C.c_int), L.ptr(src, C.c_int64), L.ptr(dst, C.c_int64), C.byref(tot)))
        assert tot.value == counts.sum()
        np.testing.assert_array_equal(src, np.arange(world) * maxn)
        np.testing.assert_array_equal(dst, np.concatenate([[0], np.cumsum(counts)[:-1]]))
    # refused: a contribution larger than the block, too small an output
    bad = np.array([3, 9], np.int32)
    with pytest.raises(L.VcError):
        parallel.gather_compact(np.zeros((2, 4, 7)), bad)


def test_frame_sharded_round_order_matches_the_compaction():
    """End to end on the CPU for world 3: chunks dealt by shard_frames, per-round contributions padded like RCCL's buffers, compacted
    by the C side -> every round's rows ascend in frame id, and concatenating the rounds reproduces the single-process row order."""
    world, chunk, T = 3, 4, 41
    all_rows = [[f + 1, 10 * f + d] for f in range(T) for d in range((f * 5) % 3)]
    shards = [parallel.shard_frames(T, r, world, chunk) for r in range(world)]
    n_rounds = (len(range(0, T, chunk)) + world - 1) // world
    out = []
    for k in range(n_rounds):
        contrib = []
        for r in range(world):
            rows = [[f + 1, 10 * f + d] for f in (range(*shards[r][k]) if k < len(shards[r]) else []) for d in range((f * 5) % 3)]
            contrib.append(np.asarray(rows, np.float64).reshape(-1, 2))
        counts = [len(c) for c in contrib]
        maxn = max(counts)
        pad = np.full((world, max(maxn, 1), 2), -1.0)
        for r, c in enumerate(contrib):
            pad[r, : len(c)] = c
        got = parallel.gather_compact(pad[:, :maxn] if maxn else pad[:, :0], counts)
        assert np.all(np.diff(got[:, 0]) >= 0)
        out.append(got)
    np.testing.assert_array_equal(np.concatenate(out, 0), np.asarray(all_rows, np.float64))


class _FakeEngine:
    def __init__(self):
        self.tuned = {}

    def tune_export(self):
        return "".join(f"{k} {v}\n" for k, v in sorted(self.tuned.items()))

    def tune_import(self, text):
        for line in text.splitlines():
            k, v = line.split()
            self.tuned[k] = int(v)


def _worker_tune(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    parallel.init_from_env("gloo")
    eng = _FakeEngine()
    warmed = []

    def warm():
        warmed.append(rank)
        eng.tuned.update({"p0_ci64_co64_k3x3": 28 + rank, "p0_ci128_co256_k1x1": 5})   # what rank 0's autotuner would have picked
    text = parallel.share_tune_cache(eng, warm)
    q.put((rank, warmed, dict(eng.tuned), text))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_tune_cache_is_tuned_once_and_broadcast():
    """Rank 0 autotunes and broadcasts its choices; rank 1 adopts them without timing anything."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker_tune, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in ps], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0] and res[1][1] == []                      # only rank 0 ran the warm-up
    assert res[0][2] == res[1][2] == {"p0_ci64_co64_k3x3": 28, "p0_ci128_co256_k1x1": 5}
    assert res[0][3] == res[1][3]
