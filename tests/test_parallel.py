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
