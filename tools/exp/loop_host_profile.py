#!/usr/bin/env python3
"""Host time of the training loop with host batches (staged ahead): how long the Python side of one iteration takes,
split into fetching the staged batch and train_step (copy into the static buffers + graph launch)."""
import argparse
import cProfile
import pstats
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from egopack_amd import data as D
from egopack_amd import engine, ops
from egopack_amd import train as T
from egopack_amd.optim import FlatAdam

T.cap_host_threads(8)
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
ops.set_compute("bf16")
order = ("ar", "lta", "oscc", "pnr")
base = {}
for t in ("ar", "lta", "pnr"):
    ds = D.SyntheticTaskDataset(t, 64, 32, 3, 1536, (115, 478), k=1, seed=1)
    base[t] = D.collate([ds[j] for j in range(64)])
hosts = []
for i in range(4):
    h = {}
    for t, b0 in base.items():
        b = D.Data(**dict(b0.__dict__))
        b.x = torch.randn(b0.x.shape, generator=torch.Generator().manual_seed(i)).to(torch.bfloat16)
        h[t] = b
    hosts.append(h)
ops.manual_seed(1000)
model, tasks, crit, weights, _, _ = bench.build_workload(args, 0, dev)
model.to(dev).train()
for t in tasks.values():
    t.to(dev).train()
opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
n = 200
it = iter(engine.StagedBatches((hosts[i % 4] for i in range(n + 10)), dev, order, fused=True))
for _ in range(8):
    b, m = next(it)
    step.train_step(b, m)
torch.cuda.synchronize()
t_fetch = t_step = 0.0
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(n):
    a = time.perf_counter()
    b, m = next(it)
    c = time.perf_counter()
    step.train_step(b, m)
    d = time.perf_counter()
    t_fetch += c - a
    t_step += d - c
pr.disable()
host = (time.perf_counter() - t0) * 1e3 / n
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3 / n
print(f"host per iteration {host:.3f} ms (fetch {t_fetch / n * 1e3:.3f}, train_step {t_step / n * 1e3:.3f}); wall incl. device {wall:.3f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
