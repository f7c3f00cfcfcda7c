"""Host time of one live training step's staging, piece by piece, without a GPU: the three resident task batches (native builder),
the merged batch, the structure keys and the packed layout (``pack_on_cpu``)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from egopack_amd import data as D  # noqa: E402
from egopack_amd.engine import structure_key  # noqa: E402

torch.set_num_threads(1)
tasks = ["ar", "lta", "pnr"]
ds = {t: D.SyntheticResidentDataset(task=t, T=32, num_segments=3, features_size=64, n_videos=8, frames=4000, length=8192, seed=1, k=1,
                                    split="train", transform=(D.LTATemporalConnectivity(r=1.5, loop=False) if t == "lta"
                                                              else D.RadiusGraph(r=1.5, loop=False))) for t in tasks}
rng = np.random.default_rng(0)
acc = {}


def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


N = 200
for it in range(N + 20):
    if it == 20:
        acc.clear()
    t0 = time.perf_counter()
    host = {t: ds[t].batch(rng.integers(0, 8192, 64)) for t in tasks}
    tick("batch x3", t0)
    t0 = time.perf_counter()
    for t in tasks:
        host[t]._struct_key = structure_key(host[t])
    tick("structure_key", t0)
    t0 = time.perf_counter()
    merged = D.merge_batches([host[t] for t in tasks])
    merged.x = None
    tick("merge_batches", t0)
    t0 = time.perf_counter()
    moved = D.to_device_packed([*(host[t] for t in tasks), merged], "cpu", pack_on_cpu=True)
    tick("to_device_packed", t0)
print("views built:", D.LazyData.fills)
print({k: round(v / N * 1e3, 3) for k, v in acc.items()}, "ms/step; sum", round(sum(acc.values()) / N * 1e3, 3))
