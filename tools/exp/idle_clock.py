import argparse, sys, time
sys.path.insert(0, ".")
import torch
import bench
from egopack_amd import engine, ops
from egopack_amd.optim import FlatAdam
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
ops.set_compute("bf16")
model, tasks, crit, weights, batches, merged = bench.build_workload(args, 0, dev)
model.to(dev).train()
for t in tasks.values(): t.to(dev).train()
opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
step.capture(batches, merged, warmup=2)
for idle_ms in (0, 1, 3, 10, 30):
    ts = []
    for _ in range(30):
        time.sleep(idle_ms / 1e3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print(f"host idle {idle_ms:3d} ms before each replay: replay + sync median {ts[len(ts)//2]:.2f} ms (min {ts[0]:.2f})")
