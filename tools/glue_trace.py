#!/usr/bin/env python3
"""Which torch-native kernels (copies, fills, adds: everything that is not an egk_* launch) does one training step
still issue, and from where?  One eager step of the bench workload under torch.profiler with python stacks; prints the
aten ops that launch device work, grouped by the innermost egopack_amd / repo frame.
Usage: python tools/glue_trace.py [--workload mtl]"""
import argparse
import collections
import sys

sys.path.insert(0, ".")
import torch
from torch.profiler import ProfilerActivity, profile

import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mtl")
a = ap.parse_args()
sys.argv = ["bench.py", "--workload", a.workload]
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload=a.workload)
dev = torch.device("cuda", 0)
from egopack_amd import engine, ops
from egopack_amd.optim import FlatAdam

ops.set_compute("bf16")
ops.manual_seed(1000)
model, tasks, crit, weights, batches, merged = bench.build_workload(args, 0, dev)
model.to(dev).train()
for t in tasks.values():
    t.to(dev).train()
params = [*model.parameters(), *(p for t in tasks.values() for p in t.parameters())]
opt = FlatAdam(params, lr=1e-5, weight_decay=1e-5)
step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True, sync=None, parallel_heads=True)
for _ in range(3):
    step.step(batches, merged)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.step(batches, merged)
    torch.cuda.synchronize()

by_site = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0:
        continue
    if ev.cpu_children and any(c.device_time_total > 0 and c.name.startswith("aten::") for c in ev.cpu_children):
        continue  # count the innermost aten op that owns the kernel
    site = "?"
    for fr in ev.stack or []:
        if "/egopack_amd/" in fr or "/root/repo" in fr or "bench.py" in fr:
            site = fr.split("/")[-1] if "/" in fr else fr
            break
    by_site[(ev.name, site)] += 1
    dur[(ev.name, site)] += ev.device_time_total
print(f"{'aten op':28s} {'n':>3s} {'us':>7s}  call site")
for k, n in sorted(by_site.items(), key=lambda kv: -dur[kv[0]]):
    print(f"{k[0]:28s} {n:3d} {dur[k]:7.1f}  {k[1]}")
print("total torch-native device time per step: %.1f us in %d launches" % (sum(dur.values()), sum(by_site.values())))
