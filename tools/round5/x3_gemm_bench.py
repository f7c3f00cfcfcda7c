#!/usr/bin/env python3
"""The precise pass's contractions (three bf16 products of f32 operands, 2048 rows) alone: the policy's launch (128-row tiles, K
split over two slabs + the reduce launch) against 64-row tiles without a split (variants 11 / 12)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev = "cuda"
for (M, N, K, two) in [(2048, 1024, 1024, False), (2048, 1024, 1024, True), (2048, 1024, 4608, False)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev)
    A2 = torch.randn(M, K, device=dev) if two else None
    B2 = torch.randn(N, K, device=dev) if two else None
    out = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    row = []
    with ops.compute_mode("bf16"), torch.no_grad(), ops.precise_scope():
        for v, sk in [(1, None), (1, 1), (11, 1), (12, 1), (3, 1), (5, 1)]:
            lib.egk_gemm_set_pipeline(v)
            kw = dict(A2=A2, lda2=K, B2=B2, ldb2=K, K2=K) if two else {}
            try:
                us = time_us(lambda: ops.gemm(M, N, A, K, B, K, K, out, N, bias=bias, compute=ops.X3, allow_splitk=sk is None, splitk=sk, **kw), 10)
                row.append(f"v{v}/sk{sk}: {us:6.1f} us")
            except Exception as e:  # noqa: BLE001
                row.append(f"v{v}/sk{sk}: {type(e).__name__}")
        lib.egk_gemm_set_pipeline(1)
    print(f"x3 {M}x{N} K={K}{' two sources' if two else ''}: " + " | ".join(row), flush=True)
