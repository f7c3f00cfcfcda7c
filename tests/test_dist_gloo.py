"""Data-parallel plumbing on CPU with the gloo backend, world_size 2 (the RCCL path on GPUs uses the
same code with backend 'nccl'): chunked all-reduce of the flat gradient buffer, parameter broadcast,
rank-sharded loaders with equal step counts."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egopack_amd import data as D
from egopack_amd.dist import GradSync, chunk_bounds, init_from_env


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sync = GradSync(world, chunk_mb=0.001)  # 262-element chunks: many chunks, reverse order
    n = 1000
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    sync.all_reduce_(g)
    ok_sum = torch.equal(g, torch.arange(n, dtype=torch.float32) * 3)  # ranks 1x + 2x
    p = torch.full((37,), float(rank + 5))
    sync.broadcast_(p)
    ok_bcast = bool((p == 5.0).all())
    # sharded loaders: disjoint samples, same number of steps
    ds = D.SyntheticTaskDataset("pnr", 21, 4, 3, 4)
    dl = D.BatchLoader(ds, 2, shuffle=True, drop_last=True, seed=7, rank=rank, world_size=world)
    steps = sum(1 for _ in dl)
    idx = dl._indices() if False else None
    gathered = [None, None]
    dist.all_gather_object(gathered, (steps, len(dl)))
    q.put((rank, ok_sum, ok_bcast, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_chunk_bounds_cover_in_reverse():
    b = chunk_bounds(10, 4)
    assert b == [(8, 10), (4, 8), (0, 4)]
    assert chunk_bounds(0, 4) == []


@pytest.mark.timeout(120)
def test_gloo_world2_allreduce_broadcast_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=100) for _ in range(2)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok_sum, ok_bcast, gathered in out:
        assert ok_sum and ok_bcast
        assert gathered[0] == gathered[1]  # identical step counts on both ranks
        assert gathered[0][0] == gathered[0][1] == 5
