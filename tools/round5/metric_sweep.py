#!/usr/bin/env python3
"""Find an operating point of tests/test_gpu_metric_target.py where the f32 validation metrics are NOT saturated (0.5-0.8):
f32 runs of main_temporal on the learnable synthetic data over (signal, classes, epochs)."""
import sys
import tempfile

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import torch

import main_temporal
from test_gpu_metric_target import COMMON, _features, _flat

for (s_ar, s_oscc, s_pnr), classes, epochs in [((0.25, 0.13, 0.10), (12, 20), 4), ((0.30, 0.12, 0.08), (12, 20), 4), ((0.35, 0.11, 0.06), (12, 20), 4),
                                              ((0.30, 0.12, 0.08), (12, 20), 6)]:
    cl = [f"dataset_{g}.num_class_labels=[{classes[0]},{classes[1]}]" for g in ("recognition", "lta", "oscc", "pnr")]
    sig = {"recognition": s_ar, "lta": s_ar, "oscc": s_oscc, "pnr": s_pnr}
    with tempfile.TemporaryDirectory() as tmp:
        args = COMMON + cl + _features(256) + [
            "batch_size=32", f"num_epochs={epochs}", "synthetic_samples=1024", "synthetic_val_samples=1024", "model.hidden_size=256",
            "model.temporal_pooling.hidden_size=256", "oscc_feat_size=256", "optimizer.lr=1e-3", f"checkpoint_dir={tmp}",
            *[f"dataset_{g}.signal={v}" for g, v in sig.items()], "compute=f32"]
        torch.manual_seed(3)
        m = _flat(main_temporal.main(args)["metrics"])
    keys = ["ar/verbs_top1", "ar/nouns_top1", "lta/verbs_top1", "lta/nouns_top1", "oscc/accuracy", "pnr/accuracy", "pnr/auroc", "pnr/recall"]
    print(f"signal {sig} classes {classes} epochs {epochs}: " + "  ".join(f"{k}={m.get(k, float('nan')):.3f}" for k in keys), flush=True)
