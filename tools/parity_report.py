#!/usr/bin/env python3
"""profiles/<tag>_config_parity.md from the figures the GPU parity tests write (gpurun_out/config_parity.jsonl by
tests/test_gpu_configs.py, gpurun_out/blockwise_parity.jsonl by tests/test_gpu_blockwise.py) in ONE run of the suite.

    python tools/parity_report.py r04 "<where the run came from>"
"""
import json
import statistics
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
origin = sys.argv[2] if len(sys.argv) > 2 else ""
rows = [json.loads(l) for l in (REPO / "gpurun_out" / "config_parity.jsonl").read_text().splitlines() if l.strip()]
blk = [json.loads(l) for l in (REPO / "gpurun_out" / "blockwise_parity.jsonl").read_text().splitlines() if l.strip()]
out = [f"# Round {tag[1:].lstrip('0')} -- parity at the full size of every BASELINE configuration (tests/test_gpu_configs.py, tests/test_gpu_blockwise.py)\n",
       f"One training step through the HIP path against the CPU oracle (oracle/path.py), measured on MI355X in the driver's collection order. {origin} "
       "All figures are relative Frobenius errors; gradients: the WORST parameter tensor. Asserted bounds (tests/test_gpu_configs.py): f32 objective <= 1e-4, "
       "gradients <= 5e-3; bf16 gradients <= 0.13 (0.27 with the OSCC head's max pool) = round-3 measurements x 1.25, k-NN lists >= 0.9995 ordered in both modes, "
       "config-4 logits <= 2.6e-2; block-wise <= 5e-3.\n"]
out.append("## f32 mode (exact-f32 MFMA, the reference's precision)\n")
out.append("| configuration | objective | loss vectors | worst gradient | Adam: fraction of parameters > 2e-4 from torch.optim.Adam |")
out.append("|---|---:|---:|---:|---:|")
for r in rows:
    if r["mode"] == "f32":
        out.append(f"| {r['config']} | {r['objective_rel']:.1e} | {r['loss_rel']:.1e} | {r['grad_rel']:.1e} ({r['grad_worst'].split('/')[-1]}) | {r.get('adam_frac_beyond_2e4', float('nan')):.1e} |")
out.append("\n## bf16 mode (the benchmark mode): three distances\n")
out.append("| configuration | loss vectors: HIP - f32 / HIP - model / model - f32 | worst gradient: HIP - f32 | HIP - model | model - f32 |")
out.append("|---|---|---:|---:|---:|")
for r in rows:
    if r["mode"] == "bf16":
        out.append(f"| {r['config']} | {r['loss_rel']:.1e} / {r.get('model_loss_rel', float('nan')):.1e} / {r.get('model_vs_f32_loss_rel', float('nan')):.1e} | "
                   f"{r['grad_rel']:.3f} | {r.get('model_grad_rel', float('nan')):.3f} | {r.get('model_vs_f32_grad_rel', float('nan')):.3f} |")
out.append("\n## bf16 mode, block by block (teacher-forced backward against the storage model)\n")
out.append("| block | worst entry | median over its tensors |")
out.append("|---|---|---:|")
ties = []
for b in blk:
    if "pairs_flipped" in b:  # the OSCC head's input gradient, stated up to near-ties of the max pool
        ties.append(b)
        continue
    vals = {k: v for k, v in b.items() if k != "block" and isinstance(v, (int, float))}
    if vals:
        k = max(vals, key=vals.get)
        out.append(f"| {b['block']} | {vals[k]:.1e} ({k}) | {statistics.median(vals.values()):.1e} |")
for b in ties:
    out.append(f"\n{b['block']}: {b['pairs_flipped']:.1e} of the (sequence, column) pairs route their gradient to another row than the model's "
               f"(each a tie within one bf16 step in the model's own features); the {b['rows_clean']:.4f} of the rows no such pair touches agree to "
               f"{b['d_features_clean_rows']:.1e}, all rows to {b['d_features_all_rows']:.1e}.")
out.append("\n## the index op (GraphONE nearest prototypes, K = 4096, k = 4, H = 1024; BASELINE #4): exact up to ties below 1e-5\n")
for r in rows:
    if r["mode"].endswith("indices"):
        out.append(f"* `{r['mode']}`: " + json.dumps({k: v for k, v in r.items() if k not in ("config", "mode")}))
(REPO / "profiles" / f"{tag}_config_parity.md").write_text("\n".join(out) + "\n")
print("\n".join(out)[:3000])
