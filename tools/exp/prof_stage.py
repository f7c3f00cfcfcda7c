import sys, time, cProfile, pstats
sys.path.insert(0, ".")
import torch
from egopack_amd import data as D, engine
order = ("ar", "lta", "oscc", "pnr")
h = {}
for t in ("ar", "lta", "pnr"):
    ds = D.SyntheticTaskDataset(t, 64, 32, 3, 1536, (115, 478), k=1, seed=1)
    b = D.collate([ds[j] for j in range(64)])
    b.x = b.x.to(torch.bfloat16)
    h[t] = b
for _ in range(3):
    engine.stage_batches(dict(h), "cuda", order)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(10):
    engine.stage_batches(dict(h), "cuda", order)
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) * 100)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
