"""Multi-process (world_size 2, gloo, CPU) test of the N>1 host path: segment sharding, barrier, max-over-ranks timing
reduction and result gathering — the same functions bench.py uses with RCCL on the GPU box."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from mgld_vsr_amd import parallel
    r, w, _ = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    n_seg = 7
    mine = parallel.shard_segments(n_seg, r, w)
    # every rank "processes" its segments: the result encodes (segment id, rank)
    frames = [torch.full((2, 3, 4, 4), float(10 * i + r)) for i in mine]
    parallel.barrier(sync_device=False)
    dt = parallel.max_over_ranks(1.0 + rank, device="cpu")
    out = parallel.gather_frames(frames, mine, n_seg)
    if rank == 0:
        ret["dt"] = dt
        ret["owners"] = [int(f[0, 0, 0, 0]) % 10 for f in out]
        ret["ids"] = [int(f[0, 0, 0, 0]) // 10 for f in out]
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_segment_sharding():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["dt"] == 2.0                                   # max over ranks
    assert ret["ids"] == list(range(7))                       # every segment processed exactly once, in order
    assert ret["owners"] == [i % world for i in range(7)]     # the reference's round-robin rule


def test_segment_bounds_and_shards():
    sys.path.insert(0, ROOT)
    from mgld_vsr_amd import parallel
    assert parallel.segment_bounds(12, 5) == [(0, 5, 0), (5, 10, 0), (10, 15, 3)]
    assert parallel.segment_bounds(10, 5) == [(0, 5, 0), (5, 10, 0)]
    got = sorted(sum((parallel.shard_segments(13, r, 4) for r in range(4)), []))
    assert got == list(range(13))
    assert parallel.shard_segments(3, 5, 8) == []
