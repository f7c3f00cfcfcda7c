#!/usr/bin/env python3
"""Soak: many replays of the captured multi-task step; the objective must stay finite and decrease, device memory must
not grow, and two runs from the same seed must end with bit-identical parameters (side streams, parallel heads, staged
graphs included).  Usage: python tools/soak.py [steps] [--staged]"""
import sys

sys.path.insert(0, ".")
import argparse

import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int, nargs="?", default=2000)
ap.add_argument("--staged", action="store_true")
a = ap.parse_args()


def run():
    from egopack_amd import engine, ops
    from egopack_amd.optim import FlatAdam
    args = argparse.Namespace(workload="mtl", batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, k=1, depth=3, compute="bf16",
                              no_fused_backbone=False, bank=4096, graphone_k=4, graphone_depth=3)
    ops.set_compute("bf16")
    ops.manual_seed(1000)
    torch.manual_seed(0)
    model, tasks, crit, weights, dev, merged = bench.build_workload(args, 0, torch.device("cuda"))
    model.cuda().train()
    for t in tasks.values():
        t.cuda().train()
    params = [*model.parameters(), *(p for t in tasks.values() for p in t.parameters())]
    opt = FlatAdam(params, lr=1e-4, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt)
    step.staged = True if a.staged else None
    step.capture(dev, merged, warmup=2)
    losses, mem0 = [], None
    for i in range(a.steps):
        total = step.replay()
        if i % 100 == 0:
            losses.append(float(total))
            if i == 100:
                mem0 = torch.cuda.memory_allocated()
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert torch.cuda.memory_allocated() <= mem0 * 1.001, (mem0, torch.cuda.memory_allocated())
    return losses, opt.flat_p.clone()


l1, p1 = run()
l2, p2 = run()
print("objective every 100 steps:", [round(v, 4) for v in l1[:12]], "...", round(l1[-1], 4))
assert l1[-1] < l1[0], "objective did not decrease"
assert l1 == l2 and torch.equal(p1, p2), "two runs from the same seed differ"
print(f"soak ok: {a.steps} replays x 2, finite, decreasing, bit-identical across runs, no memory growth")
