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

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
args = argparse.Namespace(batch=64, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl")
dev = torch.device("cuda", 0)
from egopack_amd import engine, ops
from egopack_amd.optim import FlatAdam

ops.set_compute("bf16")
for use_graph in (() if "--live" in sys.argv else (False, True)):
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


def live(workers: int, steps: int):
    """The loop of main_temporal.train with LIVE loaders: device-resident feature store, BatchLoader per task drawing fresh
    batches every step (whole-batch builders), StagedBatches one step ahead, train_step replaying the captured step."""
    from egopack_amd import data as D, train as T
    T.cap_host_threads(8)  # (as the entry points do: the training process does no arithmetic on the host)
    B, Tn = 64, 32
    order = ("ar", "lta", "oscc", "pnr")
    dsets = {t: D.SyntheticResidentDataset(t, 64 * 260, Tn, seed=1 + i, split="train", n_videos=8, frames=4000) for i, t in enumerate(("ar", "lta", "pnr"))}
    loaders = {t: D.build_dataloader(ds, B, True, 0, True, seed=1, workers=workers) for t, ds in dsets.items()}
    print(f"[loop bench] device initialised before the workers start: {torch.cuda.is_initialized()}", flush=True)
    T.start_loader_workers(loaders)  # (before the first GPU call of this process: the collation processes are a plain fork)
    ops.set_compute("bf16")
    ops.manual_seed(1000)
    model, tasks, crit, weights, _, _ = bench.build_workload(args, 0, dev)
    model.to(dev).train()
    for t in tasks.values():
        t.to(dev).train()
    opt = FlatAdam([*model.parameters(), *(p for t in tasks.values() for p in t.parameters())], lr=1e-5, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
    for ds in dsets.values():  # the tasks' datasets index ONE table (the reference's datasets read the same Omnivore file per video)
        ds.videos, ds.first_row = dsets["ar"].videos, dsets["ar"].first_row
    store = T.build_feature_store(dsets, dev)
    w = {t: (1.0 if t in dsets else 0.0) for t in order}

    # ONE pass over the loaders (leaving the iteration early would close the collation processes): the first ``warm`` steps
    # cover the eager steps, the capture and the first replays, the next ``steps`` are timed
    hosts = (dict(zip(order, b)) for b in D.multiloader([loaders.get(t) for t in order], [w[t] for t in order]))
    warm, n, t0 = 30, 0, None
    prof = None
    parts = {"replay": 0.0, "fetch": 0.0, "train_step": 0.0}
    if "--parts" in sys.argv:  # host time per step of the graph launch, the fetch of the next batch and train_step as a whole
        real_replay, real_fetch, real_ts = step.replay, engine.StagedBatches._fetch, step.train_step

        def timed(name, fn):
            def w(*a, **k):
                t = time.perf_counter()
                try:
                    return fn(*a, **k)
                finally:
                    if t0 is not None:
                        parts[name] += time.perf_counter() - t
            return w
        step.replay = timed("replay", real_replay)
        engine.StagedBatches._fetch = timed("fetch", real_fetch)
        step.train_step = timed("train_step", real_ts)
    for batches, merged in engine.StagedBatches(hosts, dev, order, fused=True, store=store, dtype=ops.act_dtype()):
        if n == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if "--profile" in sys.argv:
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
        step.train_step(batches, merged)
        n += 1
        if n == warm + steps:
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / steps
            if prof is not None:
                import pstats
                prof.disable()
                pstats.Stats(prof).sort_stats("cumtime").print_stats(40)
            if "--parts" in sys.argv:
                print("host ms / step: " + ", ".join(f"{k} {v * 1e3 / steps:.3f}" for k, v in parts.items()), flush=True)
            print(f"live loaders (workers per loader = {workers}): {ms:.3f} ms/step ({192 / ms * 1e3:.0f} clip-seqs/s), "
                  f"replayed {step.loop_counts['replayed']} / eager {step.loop_counts['eager']}", flush=True)
    for dl in loaders.values():
        dl.close()


if "--live" in sys.argv:
    for wk in [int(a.split("=")[1]) for a in sys.argv if a.startswith("--workers=")] or [0]:
        live(wk, 200)
