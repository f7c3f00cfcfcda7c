#!/usr/bin/env python3
"""The headline's grouped head contractions (AR compacted to 64 rows + LTA 2048 + PNR 2048 rows, H x H): 264 tiles of 128 rows --
the launch's tile variant by policy (96-row tiles) against 128-row tiles with one (3) / two (5) wave groups and 64-row tiles (11)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev, bf = "cuda", torch.bfloat16
for rows, tB in [((64, 2048, 2048), False), ((64, 2048, 2048), True), ((2048, 2048), False), ((64, 1920, 2048), False)]:
    probs = []
    for M in rows:
        A = torch.randn(M, 1024, device=dev).to(bf)
        B = torch.randn(1024, 1024, device=dev).to(bf)
        out = torch.empty(M, 1024, device=dev, dtype=bf)
        probs.append(((M, 1024, A, 1024, B, 1024, 1024, out, 1024), dict(transB=tB, compute=ops.BF16)))
    row = []
    for v in (1, 3, 5, 8, 11):
        lib.egk_gemm_set_pipeline(v)
        row.append(f"v{v}: {time_us(lambda: ops.gemm_grouped(probs), 20):5.1f} us")
    lib.egk_gemm_set_pipeline(1)
    print(f"rows {rows} transB={tB}: " + " | ".join(row), flush=True)
