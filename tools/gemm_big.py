#!/usr/bin/env python3
"""Contractions at the LARGE end of the workload range (config #5: 4 tasks x B=16 x T=256 = 16384 rows per GPU; B=256
sweeps; plain squares) per pipeline variant, beside the vendor library (calibration only).  Device time in a hipGraph.
Usage: python tools/gemm_big.py [variants, default 1,3,6,8]"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,3,6,8").split(",")]
dev, bf = "cuda", torch.bfloat16
SHAPES = [("fwd HxH M=16384", 16384, 1024, 1024, "nn"), ("fwd K=2H M=16384", 16384, 1024, 2048, "nn"),
          ("fwd TRN1 M=16384", 16384, 1024, 4608, "nn"), ("dX M=16384", 16384, 1024, 1024, "nt"),
          ("dW K=16384", 1024, 1024, 16384, "tt"), ("dW TRN1 K=16384", 1024, 4608, 16384, "tt"),
          ("fwd HxH M=24576", 24576, 1024, 1024, "nn"), ("sq 4096", 4096, 4096, 4096, "nn"), ("sq 8192", 8192, 8192, 8192, "nn")]
print(f"{'shape':20s} " + " ".join(f"{'v' + str(v) + ' us':>9s} {'TF/s':>5s}" for v in variants) + f" {'blas us':>9s} {'TF/s':>5s}")
for name, M, N, K, lay in SHAPES:
    if lay == "nn":
        A, B = torch.randn(M, K, device=dev).to(bf), torch.randn(N, K, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        egk = lambda: ops.gemm(M, N, A, K, B, K, K, out, N, compute=ops.BF16)
        ref = lambda: torch.matmul(A, B.t(), out=out)
    elif lay == "nt":
        A, B = torch.randn(M, K, device=dev).to(bf), torch.randn(K, N, device=dev).to(bf)
        out = torch.empty(M, N, device=dev, dtype=bf)
        egk = lambda: ops.gemm(M, N, A, K, B, N, K, out, N, transB=True, compute=ops.BF16)
        ref = lambda: torch.matmul(A, B, out=out)
    else:
        A, B = torch.randn(K, M, device=dev).to(bf), torch.randn(K, N, device=dev).to(bf)
        out = torch.zeros(M, N, device=dev)
        out16 = torch.empty(M, N, device=dev, dtype=bf)
        egk = lambda: ops.gemm(M, N, A, M, B, N, K, out, N, transA=True, transB=True, accumulate=True, compute=ops.BF16)
        ref = lambda: torch.matmul(A.t(), B, out=out16)
    fl = 2.0 * M * N * K
    cells = []
    for v in variants:
        _lib.load().egk_gemm_set_pipeline(v)
        t = time_us(egk, 5)
        cells.append(f"{t:9.1f} {fl / t / 1e6:5.0f}")
    _lib.load().egk_gemm_set_pipeline(1)
    t2 = time_us(ref, 5)
    print(f"{name:20s} " + " ".join(cells) + f" {t2:9.1f} {fl / t2 / 1e6:5.0f}", flush=True)
