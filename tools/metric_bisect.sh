#!/bin/bash
# development: which grouped path moves the bf16 training metrics (tests/test_gpu_metric_target.py, bf16 leg only)
for off in "" "wgrad_grouping" "grouped_heads" "fused_loss" "wgrad_grouping,grouped_heads,fused_loss"; do
  EGK_DISABLE=$off python - <<PY 2>/dev/null | tail -1
import sys, torch
sys.path.insert(0, ".")
import main_temporal
args = ["k=1", "seed=3", "dataset_recognition=synthetic_learnable", "dataset_lta=synthetic_learnable", "dataset_oscc=synthetic_learnable",
        "dataset_pnr=synthetic_learnable", "enabled_tasks=[ar,lta,oscc,pnr]", "save_model=False", "use_warmup=False",
        *[f"dataset_{g}.num_class_labels=[12,20]" for g in ("recognition", "lta", "oscc", "pnr")],
        *[f"dataset_{g}.features_size=256" for g in ("recognition", "lta", "oscc", "pnr")],
        *[f"dataset_{g}.signal=0.5" for g in ("recognition", "lta", "oscc", "pnr")],
        "batch_size=32", "num_epochs=6", "synthetic_samples=1024", "synthetic_val_samples=1024", "model.hidden_size=256",
        "model.temporal_pooling.hidden_size=256", "oscc_feat_size=256", "optimizer.lr=1e-3", "checkpoint_dir=/tmp/ck", "compute=bf16"]
torch.manual_seed(3)
out = main_temporal.main(args)
m = out["metrics"]
print("$off", {k: round(v, 4) for k, v in {"ar_n": m["ar"]["nouns_top1"], "lta_n": m["lta"]["nouns_top1"], "lta_loss": m["lta"]["loss"], "oscc_loss": m["oscc"]["loss"]}.items()})
PY
done
