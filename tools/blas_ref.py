#!/usr/bin/env python3
"""Calibration only (not a product path): what the vendor library (hipBLASLt through torch.matmul) reaches on the
contraction shapes of the workload, beside egk_gemm.  Device time inside a hipGraph."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import ops

dev = "cuda"
bf = torch.bfloat16
SHAPES = [  # (name, M, N, K, layout)  layout: nn = X[M,K] @ W[N,K]^T ; nt = dY[M,K] @ W[K,N] ; tt = dY[K,M]^T @ X[K,N]
    ("fwd HxH merged", 6144, 1024, 1024, "nn"), ("fwd K=2H", 6144, 1024, 2048, "nn"), ("fwd TRN1 merged", 6144, 1024, 4608, "nn"),
    ("fwd head", 2048, 1024, 1024, "nn"), ("dX merged", 6144, 1024, 1024, "nt"), ("dX head", 2048, 1024, 1024, "nt"),
    ("dW merged", 1024, 1024, 6144, "tt"), ("dW TRN1 merged", 1024, 4608, 6144, "tt"), ("dW head", 1024, 1024, 2048, "tt"),
    ("sq 4096", 4096, 4096, 4096, "nn"), ("sq 8192", 8192, 8192, 8192, "nn"),
]
print(f"{'shape':20s} {'M':>5s} {'N':>5s} {'K':>5s}  {'egk us':>8s} {'TF/s':>6s}  {'blas us':>8s} {'TF/s':>6s}")
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
    t1, t2 = time_us(egk, 10), time_us(ref, 10)
    print(f"{name:20s} {M:5d} {N:5d} {K:5d}  {t1:8.1f} {fl / t1 / 1e6:6.0f}  {t2:8.1f} {fl / t2 / 1e6:6.0f}", flush=True)
