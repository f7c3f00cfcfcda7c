#!/usr/bin/env python3
"""Launch ONE contraction shape a few times (for rocprofv3 --pmc passes).
Usage: python3 tools/gemm_one.py M N K tA tB [splitk] [pipeline]"""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import _lib, ops

M, N, K, tA, tB = (int(v) for v in sys.argv[1:6])
sk = int(sys.argv[6]) if len(sys.argv) > 6 else None
if len(sys.argv) > 7:
    _lib.load().egk_gemm_set_pipeline(int(sys.argv[7]))
dt = torch.bfloat16
A = torch.randn((K, M) if tA else (M, K), device="cuda").to(dt)
B = torch.randn((K, N) if tB else (N, K), device="cuda").to(dt)
acc = bool(tA and tB)
out = torch.zeros(M, N, device="cuda", dtype=torch.float32 if acc else dt)
for _ in range(5):
    ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, out, N, transA=bool(tA), transB=bool(tB), accumulate=acc,
             compute=ops.BF16, splitk=sk)
torch.cuda.synchronize()
