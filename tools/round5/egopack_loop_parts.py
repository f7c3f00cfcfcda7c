#!/usr/bin/env python3
"""Host time per step of main_egopack.py's loop by part (fetch of the next batch, train_step, the graph launch inside it) at the
benchmark's shapes.  Needs the checkpoint tools/round5/entry_loops.sh's first phase leaves in /tmp/ck (same box)."""
import sys
import time

sys.path.insert(0, ".")
import torch

from egopack_amd import engine

parts, on = {"replay": 0.0, "fetch": 0.0, "train_step": 0.0, "steps": 0}, {"t": False}


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            if on["t"]:
                parts[name] += time.perf_counter() - t
    return w


engine.StagedBatches._fetch = timed("fetch", engine.StagedBatches._fetch)
real_ts = engine.StepBase.train_step


def ts(self, *a, **k):
    parts["steps"] += 1
    if parts["steps"] == 30:
        torch.cuda.synchronize()
        on["t"], parts["t0"] = True, time.perf_counter()
        if not getattr(self, "_timed_replay", False):
            self.replay, self._timed_replay = timed("replay", self.replay), True
            g = self._graph
            parts["graph"] = 0.0
            print("[parts] graph object:", type(g).__name__, "notes:", getattr(self, "capture_notes", None), "one_pass:",
                  getattr(self, "one_pass", None), "hyper_in_graph:", getattr(self, "_hyper_in_graph", None), flush=True)
            g_replay = g.replay
            try:
                g.replay = timed("graph", g_replay)
            except Exception as e:  # noqa: BLE001
                print("[parts] cannot wrap graph.replay:", e)
            self.optimizer.sync_hyper_source = timed("sync_hyper", self.optimizer.sync_hyper_source)
            parts["sync_hyper"] = 0.0
    return timed("train_step", real_ts)(self, *a, **k)


engine.StepBase.train_step = ts
import main_egopack

G = "dataset_recognition=synthetic_resident dataset_lta=synthetic_resident dataset_oscc=synthetic_resident dataset_pnr=synthetic_resident".split()
S = [f"dataset_{d}.{k}={v}" for d in ("recognition", "lta", "oscc", "pnr") for k, v in (("T", 32), ("n_videos", 8), ("frames", 4000))]
C = ("k=1 batch_size=64 synthetic_samples=8192 synthetic_val_samples=64 model.hidden_size=1024 model.temporal_pooling.hidden_size=1024 "
     "compute=bf16 checkpoint_dir=/tmp/ck num_epochs=1 enabled_tasks=[oscc] enable_graphone=True "
     "resume_from=/tmp/ck/MTL_ar-lta-pnr/checkpoint.pth graphone.k=4 graphone.depth=3 graphone.residual=True save_model=False").split()
main_egopack.main(G + S + C)
torch.cuda.synchronize()
n = parts["steps"] - 30
wall = (time.perf_counter() - parts["t0"]) * 1e3 / max(n, 1)
print(f"[parts] {n} steps: " + ", ".join(f"{k} {parts[k] * 1e3 / n:.3f}" for k in ("fetch", "train_step", "replay", "graph", "sync_hyper") if k in parts) + f" ms/step of host time; wall <= {wall:.3f} ms/step (incl. the epoch's end)", flush=True)
