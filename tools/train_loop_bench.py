#!/usr/bin/env python3
"""What a training LOOP gets per step (engine.StepBase.train_step on device batches of the bench workload, fresh values
copied in every step): the eager step against the captured step with its per-step value copies.
Usage: python tools/train_loop_bench.py [steps]"""
import argparse
import sys
import time

sys.path.insert(0, ".")
import torch

import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
dev = torch.device("cuda", 0)
from egopack_amd import engine, ops
from egopack_amd.optim import FlatAdam

ops.set_compute("bf16")
for use_graph in (False, True):
    ops.manual_seed(1000)
    model, tasks, crit, weights, batches, merged = bench.build_workload(args, 0, dev)
    model.to(dev).train()
    for t in tasks.values():
        t.to(dev).train()
    opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
    step.use_graph = use_graph
    # (what stage_batches attaches to the batches of a loader: equal structure fingerprints -> the replay path rewrites
    #  features, labels and positions only)
    for i, b in enumerate([*batches.values(), merged]):
        b._struct_key = 12345 + i
    for _ in range(5):
        step.train_step(batches, merged)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.train_step(batches, merged)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    print(f"train_step, use_graph={use_graph}: {ms:.3f} ms/step ({192 / ms * 1e3:.0f} clip-seqs/s)")
