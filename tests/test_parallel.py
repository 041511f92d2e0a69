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
