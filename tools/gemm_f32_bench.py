#!/usr/bin/env python3
"""Exact-f32 contractions (the reference-precision mode): the LDS-DMA pipelined kernel against the register-staged generic
kernel on the shapes of the headline step and on squares, device time per launch inside a hipGraph.
Usage: python tools/gemm_f32_bench.py [--iters 10]"""
import argparse
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import _lib, ops
from tools._timing import time_us

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
args = ap.parse_args()
N6, H = 6144, 1024
SHAPES = [("fwd HxH merged", N6, H, H, False, False), ("fwd SAGE combine K=2H", N6, H, 2 * H, False, False),
          ("fwd TRN1 merged", N6, H, 4608, False, False), ("fwd head", 2048, H, H, False, False),
          ("dX HxH merged", N6, H, H, False, True), ("dX TRN1", N6, 4608, H, False, True),
          ("dW HxH merged", H, H, N6, True, True), ("dW TRN1 merged", H, 4608, N6, True, True), ("dW head", H, H, 2048, True, True),
          ("NN 4096^3", 4096, 4096, 4096, False, False), ("NT 4096^3", 4096, 4096, 4096, False, True),
          ("TT 4096^3", 4096, 4096, 4096, True, True), ("TN 4096^3", 4096, 4096, 4096, True, False)]
lib = _lib.load()
print(f"{'shape':26s} {'M':>5s} {'N':>5s} {'K':>5s} splitk  {'generic us':>10s} {'TF/s':>6s}  {'pipelined us':>12s} {'TF/s':>6s}  frac of 157.3")
for name, M, N, K, tA, tB in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device="cuda")
    B = torch.randn((K, N) if tB else (N, K), device="cuda")
    out = torch.zeros(M, N, device="cuda")
    acc = tA and tB
    sk = lib.egk_gemm_splitk(M, N, K, ops.F32)

    def run():
        ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, out, N, transA=tA, transB=tB, accumulate=acc, compute=ops.F32)
    cells = []
    for v in (0, 1):
        lib.egk_gemm_set_pipeline(v)
        us = time_us(run, args.iters)
        cells.append((us, 2.0 * M * N * K / us / 1e6))
    lib.egk_gemm_set_pipeline(1)
    print(f"{name:26s} {M:5d} {N:5d} {K:5d} {sk:6d}  {cells[0][0]:10.1f} {cells[0][1]:6.1f}  {cells[1][0]:12.1f} {cells[1][1]:6.1f}  {cells[1][1] / 157.3:.2f}")
