#!/usr/bin/env python3
"""Copy a tools/profile.sh output (gpurun_out/prof_<tag>/) into profiles/ under a round tag and stamp the PMC summary with
its provenance: the source commit the numbers were measured at (this script runs in the build container, where .git is).

    python tools/stamp_profile.py r03a            # gpurun_out/prof_r03a -> profiles/r03a_*, profiles/pmc_latest.json
    python tools/stamp_profile.py r04_c4 egopack_oscc_B64_T32_H1024_Hp1024_bf16
                                                  # ... -> profiles/r04_c4_*, profiles/pmc_<key>.json (bench.pmc_key of that run)
"""
import datetime
import json
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else None
src = REPO / "gpurun_out" / f"prof_{tag}"
dst = REPO / "profiles"
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=REPO, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "egopack_amd", "bench.py"], cwd=REPO, capture_output=True,
                            text=True).stdout.strip())
if len(sys.argv) > 3:  # the commit the GPU run was taken at, when the tree has moved on since
    commit, dirty = sys.argv[3], False
meta = {"commit": commit + ("+uncommitted" if dirty else ""), "tag": tag,
        "date": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M UTC")}
copied = []
for name, out in (("summary.md", f"{tag}_rocprofv3_summary.md"), ("pmc.json", f"{tag}_pmc.json"), ("mfma.json", f"{tag}_mfma.json")):
    f = src / name
    if f.exists():
        if name.endswith(".json"):
            d = json.loads(f.read_text())
            d["_meta"] = meta
            (dst / out).write_text(json.dumps(d, indent=1))
            if name == "pmc.json":
                d["_meta"]["config"] = key or "mtl_B64_T32_H1024_Hp1024_bf16"
                (dst / (f"pmc_{key}.json" if key else "pmc_latest.json")).write_text(json.dumps(d, indent=1))
        else:
            text = f.read_text()
            (dst / out).write_text(text + f"\n\n(measured at source commit {meta['commit']}, {meta['date']})\n")
        copied.append(out)
stats = sorted((src / "stats").rglob("*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst / f"{tag}_kernel_stats.csv")
    copied.append(f"{tag}_kernel_stats.csv")
print("profiles/:", ", ".join(copied), "| meta", meta)
