#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel stats (rocprofv3 --kernel-trace --stats)
and HBM traffic per launch from the PMC passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE; KiB units),
as prescribed by MI355X_MICROARCH.md (HBM section)."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])


def find(sub, pat):
    hits = sorted((root / sub).rglob(pat))
    return hits[0] if hits else None


def short(name):
    name = name.replace("void egk::", "").replace("egk::", "")
    return name[:90]


print(f"# rocprofv3 summary ({root.name})\n")
stats = find("stats", "*kernel_stats.csv")
if stats:
    print("## kernel stats (--kernel-trace --stats), whole bench.py run\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---:|---:|---:|---:|")
    with open(stats) as f:
        rows = list(csv.DictReader(f))
    # bench.py's live kernel timing is taken behind a ~35 ms device-side spin (at::cuda::spin_kernel, one launch): it is
    # not part of the workload -- percentages below are over everything else
    spin = [r for r in rows if "spin_kernel" in r["Name"]]
    rows = [r for r in rows if "spin_kernel" not in r["Name"]]
    total = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
    for r in rows[:40]:
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.2f} | {100.0 * float(r['TotalDurationNs']) / total:.2f} |")
    if spin:
        print(f"\n(excluded: {spin[0]['Calls']} x `spin_kernel`, {float(spin[0]['TotalDurationNs']) / 1e6:.1f} ms -- the spin bench.py keeps "
              "the device queue full with while it takes its live per-kernel timings)")
    print()

import json
pmc = {}
for sub, ctr, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    agg = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != ctr:
                continue
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    print(f"## {ctr} per launch (KiB x 1024 x {mult:g} = bytes; gfx950 correction per the microarch guide)\n")
    print("| kernel | launches | MB per launch |")
    print("|---|---:|---:|")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"| {short(k)} | {n} | {v * 1024 * mult / n / 1e6:.2f} |")
    print()
    for k, (n, v) in agg.items():
        pmc.setdefault(short(k), {})[ctr] = {"launches": n, "bytes_per_launch": v * 1024 * mult / n}
# matrix-pipe counters (pmc_mfma pass): per kernel, summed over its launches
f = find("pmc_mfma", "*counter_collection.csv")
if f:
    agg = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                calls[r["Kernel_Name"]] += 1
    print("## matrix-pipe utilisation (SQ counters, own pass)\n")
    print("MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32).  Calibration on this chip: "
          "SQ_VALU_MFMA_BUSY_CYCLES is the matrix-pipe cycles summed over all 1024 SIMDs (6144 x 1024 x 1024 bf16 = 786 432 "
          "v_mfma_f32_16x16x32_bf16 x 16 cycles = 12 582 912, the value reported), SQ_BUSY_CYCLES is summed over the 32 shader "
          "engines (32 SIMDs each), so the quotient is the fraction of the kernel's busy SIMD-cycles in which the matrix pipe "
          "was executing.  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers), issue-stall = "
          "SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES.\n")
    print("| kernel | launches | MFMA pipe utilisation | wait / wave cycles | issue stall / wave cycles | bf16 MOPS per launch |")
    print("|---|---:|---:|---:|---:|---:|")
    mfma = {}
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0))[:25]:
        busy, wave = c.get("SQ_BUSY_CYCLES", 0.0) or 1.0, c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        n = max(calls[k], 1)
        row = {"launches": n, "mfma_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (busy * 32.0),
               "wait_frac": c.get("SQ_WAIT_ANY", 0.0) / wave, "issue_stall_frac": c.get("SQ_WAIT_INST_ANY", 0.0) / wave,
               "mops_bf16_per_launch": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) / n}
        mfma[short(k)] = row
        print(f"| {short(k)} | {n} | {row['mfma_busy_frac']:.3f} | {row['wait_frac']:.3f} | {row['issue_stall_frac']:.3f} | "
              f"{row['mops_bf16_per_launch']:.3g} |")
    print()
    (root / "mfma.json").write_text(json.dumps(mfma, indent=1))
(root / "pmc.json").write_text(json.dumps(pmc, indent=1))
