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
h = {}
for t in ("ar", "lta", "pnr"):
    ds = D.SyntheticTaskDataset(t, 64, 32, 3, 1536, (115, 478), k=1, seed=1)
    b = D.collate([ds[j] for j in range(64)]); b.x = b.x.to(torch.bfloat16); h[t] = b
model, tasks, crit, weights, _, _ = bench.build_workload(args, 0, dev)
model.to(dev).train()
for t in tasks.values(): t.to(dev).train()
opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
acc = {"stage": 0.0, "sig": 0.0, "copy": 0.0, "replay": 0.0}
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    batches, merged = engine.stage_batches(dict(h), dev, order)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if i < 6:
        step.train_step(batches, merged); continue
    sig = engine.batch_signature(batches, merged)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = step._train_static
    engine.copy_batch_values(st["batches"], st["merged"], batches, merged)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    step.replay()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): acc[k] += v
    if i % 4 == 0: print(i, [round(x * 1e3, 2) for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3)])
print({k: round(v / 34 * 1e3, 2) for k, v in acc.items()}, "ms per step")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step.replay()
torch.cuda.synchronize()
print("back-to-back replays of the train_step graph:", round((time.perf_counter() - t0) * 50, 3), "ms each")
import collections
print("static merged x:", st["merged"].x.shape, st["merged"].x.dtype, "graph rows", st["merged"].graph.rowptr.shape, "heavy", st["merged"].graph.t_heavy.numel(), st["merged"].graph.t_heavy_mode)
pin = torch.empty(57 * 1024 * 1024, dtype=torch.uint8, pin_memory=True)
dst = torch.empty_like(pin, device=dev)
def timed(label, pre):
    ts = []
    for _ in range(15):
        pre()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step.replay()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); print(f"{label:48s} replay median {ts[7]:.2f} ms")
timed("nothing before", lambda: None)
timed("57 MB pinned H2D before", lambda: dst.copy_(pin, non_blocking=True))
timed("stage_batches before", lambda: engine.stage_batches(dict(h), dev, order))
def stage_and_copy():
    b, m = engine.stage_batches(dict(h), dev, order)
    engine.copy_batch_values(st["batches"], st["merged"], b, m)
timed("stage_batches + copy_batch_values before", stage_and_copy)
timed("host busy 20 ms (python loop) before", lambda: sum(i * i for i in range(400000)))
