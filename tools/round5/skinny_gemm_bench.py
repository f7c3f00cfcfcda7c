#!/usr/bin/env python3
"""64-row contractions (the compacted AR head of BASELINE config 2): the policy's split + reduce against one launch without a split."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev, bf = "cuda", torch.bfloat16
for (M, N, K, tB) in [(64, 1024, 1024, False), (64, 1024, 1024, True), (64, 640, 1024, False), (128, 1024, 1024, False), (64, 1024, 2048, False)]:
    A = torch.randn(M, K, device=dev).to(bf)
    B = (torch.randn(K, N, device=dev) if tB else torch.randn(N, K, device=dev)).to(bf)
    out = torch.empty(M, N, device=dev, dtype=bf)
    row = []
    for sk in (None, 1, 2, 4):
        us = time_us(lambda: ops.gemm(M, N, A, K, B, B.shape[1], K, out, N, transB=tB, compute=ops.BF16, allow_splitk=sk is None, splitk=sk), 20)
        row.append(f"sk{sk}: {us:5.1f} us")
    print(f"{M}x{N} K={K} transB={tB}: auto split = {lib.egk_gemm_splitk(M, N, K, ops.BF16)} | " + " | ".join(row), flush=True)
