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
    st = torch.tensor([[1.0, 2.0, 10.0], [3.0, 4.0, 20.0]], dtype=torch.float64) * (rank + 1)  # graph-LN sums + counts
    sync.sum_small(st)
    ok_sum = ok_sum and torch.equal(st, torch.tensor([[3.0, 6.0, 30.0], [9.0, 12.0, 60.0]], dtype=torch.float64))
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


# ---- evaluation passes over several ranks (SURVEY 8(e) caveats 3 and 5) -------------------------------------------
class _DS:
    """What the meters read from a dataset."""
    label_names = ["verbs", "nouns"]
    class_labels = [[f"v{i}" for i in range(5)], [f"n{i}" for i in range(7)]]
    lta_nodes = 6
    task = "ar"


def _fake_meters(seed):
    """One meter of every kind on the CPU with a seeded, plausible state (the kernels that fill them need a GPU; the
    cross-rank combination is host logic over the state tensors)."""
    from egopack_amd import meters as M
    g = torch.Generator().manual_seed(seed)
    ri = lambda hi, *shape: torch.randint(0, hi, shape, generator=g)
    out = []
    for cls in (M.RecognitionMeter, M.AnticipationMeter, M.LTAMeter):
        m = cls(_DS(), device="cpu")
        for h in (m.verbs, m.nouns):
            for t in h.tensors():
                t.copy_(ri(50, *t.shape) if t.dim() else ri(50, 1)[0])
        if isinstance(m, M.LTAMeter):
            m.ed_sum += torch.rand(2, generator=g, dtype=torch.float64)
            m.ed_n = int(ri(20, 1)) + 1
        out.append(m)
    o = M.OSCCMeter(device="cpu")
    for t in o.counts.tensors():
        t.copy_(ri(50, *t.shape) if t.dim() else ri(50, 1)[0])
    out.append(o)
    p = M.PNRMeter(device="cpu")
    p.stats += ri(30, 4)
    n = 5 + seed  # ranks hold different numbers of scores (padded all-gather)
    p.probs, p.targets = [torch.rand(n, generator=g)], [ri(2, n).bool()]
    p.targets[0][0], p.targets[0][1] = True, False
    p.loc_err_sum += torch.rand((), generator=g, dtype=torch.float64)
    p.loc_n = 3 + seed
    out.append(p)
    for m in out:
        m.loss_sum += torch.rand((), generator=g, dtype=torch.float64)
        m.loss_n, m.counter = 2 + seed, 10 + seed
    return out


def _scalars(m):
    return {k: v for k, v in m.get_logs().items() if isinstance(v, (int, float))}


def _eval_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    init_from_env(backend="gloo")
    logs = [_scalars(m.all_reduce()) for m in _fake_meters(rank)]
    ds = D.SyntheticTaskDataset("pnr", 23, 4, 3, 4)
    dl = D.BatchLoader(ds, 4, shuffle=False, drop_last=False, rank=rank, world_size=world, shard="batches")
    mine = [b.x.clone() for b in dl]
    assert len(mine) == len(dl)
    q.put((rank, logs, [m.tolist() for m in mine]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_meters_and_batch_sharded_eval_loader():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=100) for _ in range(2)), key=lambda o: o[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # the in-process merge of the two ranks' meters is the expected total
    want = [_scalars(a.merge(b)) for a, b in zip(_fake_meters(0), _fake_meters(1))]
    for rank, logs, _ in out:
        for got, exp in zip(logs, want):
            assert got.keys() == exp.keys()
            for k in exp:
                assert got[k] == pytest.approx(exp[k], rel=1e-12, abs=1e-12), (rank, k)
    # the two ranks' batches interleave into exactly the single-process batches (same rows, same batch boundaries)
    ds = D.SyntheticTaskDataset("pnr", 23, 4, 3, 4)
    single = [b.x.tolist() for b in D.BatchLoader(ds, 4, shuffle=False, drop_last=False)]
    assert len(single) == 6 and len(out[0][2]) == 3 and len(out[1][2]) == 3
    assert [out[i % 2][2][i // 2] for i in range(6)] == single


# ---- sharded update: reduce-scatter -> Adam on this rank's slice -> all-gather (VERDICT r2 #5b) -----------------------------
class _CpuAdam:
    """The slice-wise interface of optim.FlatAdam (flat buffers, prepare_hyper, launch(grads, lo, hi), refresh_shadows) in
    plain torch on the CPU: what dist.GradSync drives.  Test infrastructure (the product optimizer needs a ROCm device)."""

    def __init__(self, n):
        g = torch.Generator().manual_seed(3)
        self.flat_p = torch.randn(n, generator=g)
        self.flat_g, self.flat_m, self.flat_v = torch.zeros(n), torch.zeros(n), torch.zeros(n)
        self.step_count, self.grad_scale, self.refreshed = 0, 1.0, 0

    def prepare_hyper(self):
        t = self.step_count + 1
        self._c = (1.0 - 0.9 ** t, (1.0 - 0.999 ** t) ** 0.5, self.grad_scale)

    def launch(self, grads=None, lo=0, hi=None):
        sl = slice(lo, self.flat_p.numel() if hi is None else hi)
        g = (self.flat_g if grads is None else grads)[sl] * self._c[2]
        self.flat_m[sl] = 0.9 * self.flat_m[sl] + 0.1 * g
        self.flat_v[sl] = 0.999 * self.flat_v[sl] + 0.001 * g * g
        self.flat_p[sl] -= 1e-2 * (self.flat_m[sl] / self._c[0]) / (self.flat_v[sl].sqrt() / self._c[1] + 1e-8)

    def refresh_shadows(self):
        self.refreshed += 1


def _sharded_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    init_from_env(backend="gloo")
    n = 1000  # slices of 496 elements, an 8-element remainder stepped on both ranks
    ref, shd = _CpuAdam(n), _CpuAdam(n)
    sync = GradSync(world, shard_update=True)
    per, lo, hi, body = sync.shard_bounds(n)
    ok = (per, lo, hi, body) == (496, rank * 496, (rank + 1) * 496, 992)
    for step in range(3):
        g = torch.randn(n, generator=torch.Generator().manual_seed(100 * step + rank))
        # all-reduce, then the whole Adam pass on every rank
        ref.flat_g.copy_(g)
        dist.all_reduce(ref.flat_g)
        ref.grad_scale = 1.0 / world
        ref.prepare_hyper()
        ref.launch()
        ref.step_count += 1
        # reduce-scatter (all-reduce on gloo), Adam on the own slice + the remainder, all-gather of the parameters
        shd.flat_g.copy_(g)
        sync.reduce_and_step(shd)
    ok = ok and torch.equal(ref.flat_p, shd.flat_p) and shd.step_count == 3 and shd.refreshed == 3
    other = torch.ones(n, dtype=torch.bool)
    other[lo:hi] = False
    other[body:] = False
    ok = ok and not shd.flat_m[other].any() and bool(shd.flat_m[lo:hi].any()) and torch.equal(shd.flat_m[lo:hi], ref.flat_m[lo:hi])
    # a checkpoint needs the FULL moments: the sharded step flags the optimizer, gather_moments (a collective) rebuilds them
    ok = ok and getattr(shd, "_moments_sharded", False) is True
    sync.gather_moments(shd)
    ok = ok and shd._moments_sharded is False and torch.equal(shd.flat_m, ref.flat_m) and torch.equal(shd.flat_v, ref.flat_v)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_sharded_update_equals_allreduce_then_full_adam():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]


# ---- bench.py's pre-flight probe (--exchange-graph auto): the lock-step polling of the rank processes, with stand-in children ----
def _probe_worker(rank, world, port, q, cmds, timeout):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import argparse
    import time as _t
    import bench
    init_from_env(backend="gloo")
    out = []
    for cmd in cmds:
        os.environ["EGK_TEST_PROBE_CMD"] = cmd[rank]
        t0 = _t.monotonic()
        ok, note = bench.one_graph_probe(argparse.Namespace(probe_timeout=timeout), [], rank, world)
        out.append((bool(ok), note, round(_t.monotonic() - t0, 1)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_probe_verdict_is_the_same_on_every_rank():
    """``bench.one_graph_probe``: every rank starts a child and the ranks poll in lock step over the gloo group.  Both children
    exit 0 -> every rank takes the one-graph mode; ONE child fails (exit code 3 on rank 1) -> every rank takes the staged
    graphs, at once; a child that hangs is killed at the timeout and counts as failed -- on both ranks, with the other rank's
    (still running) child killed too."""
    import sys
    py = sys.executable
    ok = f'{py} -c "import sys; sys.exit(0)"'
    bad = f'{py} -c "import sys; sys.exit(3)"'
    slow_ok = f'{py} -c "import time; time.sleep(1.5)"'
    hang = f'{py} -c "import time; time.sleep(600)"'
    cmds = [(ok, slow_ok), (slow_ok, bad), (hang, ok)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, 2, port, q, cmds, 4.0)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(30)
    for r in (0, 1):
        (ok1, note1, _), (ok2, note2, t2), (ok3, note3, t3) = res[r]
        assert ok1 and note1 == "passed", res
        assert not ok2 and ("exit code 3" in note2 or "another rank" in note2) and t2 < 3.5, res
        assert not ok3 and ("timed out" in note3 or "another rank" in note3) and t3 < 8.0, res


# ---- region-wise sharded update: reduce-scatter per region as backward finishes it (VERDICT r5 #1b) ---------------------------------
def _region_sharded_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_from_env(backend="gloo")
    n = 2008
    regions = [(1200, 2008), (400, 1200), (0, 400)]  # heads, SAGE stack, TRN: the order backward finishes them in
    ref, shd = _CpuAdam(n), _CpuAdam(n)
    calls = []
    shd.refresh_shadows = lambda lo=0, hi=None: calls.append((lo, hi))
    sync = GradSync(world, shard_update=True)
    ok = sync.shard_bounds(808, 1200) == (400, 1200 + rank * 400, 1200 + (rank + 1) * 400, 2000)
    for step in range(3):
        g = torch.randn(n, generator=torch.Generator().manual_seed(100 * step + rank))
        ref.flat_g.copy_(g)
        dist.all_reduce(ref.flat_g)
        ref.grad_scale = 1.0 / world
        ref.prepare_hyper()
        ref.launch()
        ref.step_count += 1
        shd.flat_g.copy_(g)
        sync.begin_step()
        for lo, hi in regions:
            sync.start(shd, lo, hi)  # (reduce-scatter of the region; the engine calls this between the backward stages)
        sync.finish_and_step(shd)    # Adam on the own slices, all-gather of the parameters, shadows of the gathered slices
    ok = ok and torch.equal(ref.flat_p, shd.flat_p) and shd.step_count == 3
    ok = ok and calls[:3] == [(1200, 2000), (400, 1200), (0, 400)]
    mine = torch.zeros(n, dtype=torch.bool)
    for lo, hi in regions:
        per, a, b, body = sync.shard_bounds(hi - lo, lo)
        mine[a:b] = True
        mine[body:hi] = True
    ok = ok and not shd.flat_m[~mine].any() and torch.equal(shd.flat_m[mine], ref.flat_m[mine])
    ok = ok and getattr(shd, "_moments_sharded", False) is True
    sync.gather_moments(shd)
    ok = ok and shd._moments_sharded is False and torch.equal(shd.flat_m, ref.flat_m) and torch.equal(shd.flat_v, ref.flat_v)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_region_wise_sharded_update_equals_allreduce_then_full_adam():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_region_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]
