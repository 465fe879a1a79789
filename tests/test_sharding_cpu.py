"""Multi-process path on CPU (gloo, world_size 2): frames are sharded by rank
with no data-path collective; the only communication bench.py performs is the
barrier + MAX-reduce of the elapsed time, exercised here with gloo."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import pointgnn_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_frames, q):
    sys.path.insert(0, ROOT)
    import pointgnn_amd  # noqa: F401
    from pointgnn_amd.engine import shard_frames
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_frames(num_frames, rank, world)
    # every rank "processes" its frames: here, a checksum of the frame ids
    local = torch.tensor([float(sum(mine)), float(len(mine))], dtype=torch.float64)
    dist.barrier()
    elapsed = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)     # bench.py's timing rule
    gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, local)
    q.put((rank, mine, float(elapsed.item()), [g.tolist() for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_frames_partition():
    from pointgnn_amd.engine import shard_frames
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 16, 100):
            parts = [shard_frames(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))                 # disjoint cover
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_frames(4, 2, 2)


def test_two_process_gloo_sharding():
    world, n = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    frames = sorted(i for _, mine, _, _ in results for i in mine)
    assert frames == list(range(n))
    for rank, mine, elapsed, gathered in results:
        assert elapsed == 1.5                              # MAX over ranks
        assert sum(g[1] for g in gathered) == n
        assert sum(g[0] for g in gathered) == sum(range(n))
