#!/usr/bin/env python3
"""Training loop with host batches (features cross PCIe every step): staging on the compute stream against
engine.StagedBatches (next batch collated and copied on a copy stream while the step runs).  The host batches are
pre-collated (the synthetic dataset's per-sample randn would dominate otherwise).  Usage: python tools/loop_staging_bench.py"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from egopack_amd import data as D
from egopack_amd import engine, ops
from egopack_amd.optim import FlatAdam

from egopack_amd import train as T

T.cap_host_threads(8)  # (what the entry points do: see train.cap_host_threads)
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
ops.set_compute("bf16")
order = ("ar", "lta", "oscc", "pnr")
hosts = []
base = {}
for t in ("ar", "lta", "pnr"):
    ds = D.SyntheticTaskDataset(t, 64, 32, 3, 1536, (115, 478), k=1, seed=1)
    base[t] = D.collate([ds[j] for j in range(64)])
for i in range(4):  # four host steps with their own features, cycled (same sequence lengths: the structure a loader of
    h = {}          # fixed-length sequences delivers every step)
    for t, b0 in base.items():
        b = D.Data(**dict(b0.__dict__))
        b.x = torch.randn(b0.x.shape, generator=torch.Generator().manual_seed(i)).to(torch.bfloat16)
        h[t] = b
    hosts.append(h)
for mode in ("inline", "ahead"):
    ops.manual_seed(1000)
    model, tasks, crit, weights, _, _ = bench.build_workload(args, 0, dev)
    model.to(dev).train()
    for t in tasks.values():
        t.to(dev).train()
    opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
    n = 150
    stream_of_hosts = (hosts[i % 4] for i in range(n + 6))
    it = 0
    t0 = None
    if mode == "inline":
        for h in stream_of_hosts:
            batches, merged = engine.stage_batches(dict(h), dev, order)
            step.train_step(batches, merged)
            it += 1
            if it == 6:
                torch.cuda.synchronize(); t0 = time.perf_counter()
    else:
        for batches, merged in engine.StagedBatches(stream_of_hosts, dev, order, fused=True):
            step.train_step(batches, merged)
            it += 1
            if it == 6:
                torch.cuda.synchronize(); t0 = time.perf_counter()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    print(f"host batches, staging {mode:6s}: {ms:.3f} ms/step ({192 / ms * 1e3:.0f} clip-seqs/s)")
