import argparse, sys, time
sys.path.insert(0, ".")
import torch
import bench
from egopack_amd import data as D, engine, ops
from egopack_amd.optim import FlatAdam
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
ops.set_compute("bf16")
order = ("ar", "lta", "oscc", "pnr")
base = {}
for t in ("ar", "lta", "pnr"):
    ds = D.SyntheticTaskDataset(t, 64, 32, 3, 1536, (115, 478), k=1, seed=1)
    base[t] = D.collate([ds[j] for j in range(64)])
    base[t].x = base[t].x.to(torch.bfloat16)
model, tasks, crit, weights, _, _ = bench.build_workload(args, 0, dev)
model.to(dev).train()
for t in tasks.values(): t.to(dev).train()
opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
rows = []
for i in range(60):
    t0 = time.perf_counter()
    batches, merged = engine.stage_batches(dict(base), dev, order)
    t1 = time.perf_counter()
    step.train_step(batches, merged)
    t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1))
torch.cuda.synchronize()
for i in range(0, 60, 4):
    print(i, [round(x * 1e3, 2) for x in rows[i]])
for nt in (128, 16, 4):
    torch.set_num_threads(nt)
    ts = []
    for i in range(60):
        t0 = time.perf_counter()
        batches, merged = engine.stage_batches(dict(base), dev, order)
        step.train_step(batches, merged)
        ts.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    ts2 = sorted(ts)
    print(f"threads {nt:3d}: per-step host time median {ts2[30]:.2f} ms, mean {sum(ts)/60:.2f} ms, max {ts2[-1]:.1f} ms, >20ms: {sum(t > 20 for t in ts)}")
