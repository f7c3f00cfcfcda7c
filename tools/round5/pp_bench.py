#!/usr/bin/env python3
"""The 256 x 128 ping-pong kernel (egk_gemm_set_pipeline(14)) against the 128 x 128 two-workgroups-per-CU kernel (3), the lock-step
256 x 128 tile (6 / 13) and the 256 x 256 tile (7) on the large contractions of the Hp = 4096 temporal pooling and on squares."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev, bf = "cuda", torch.bfloat16
variants = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,6,13,14,7,15,1").split(",")]
SHAPES = [("fwd TRN1 Hp4096", 6144, 4096, 4608, False, False), ("fwd TRN2 Hp4096", 6144, 4096, 4096, False, False),
          ("dX TRN2 Hp4096", 6144, 4096, 4096, False, True), ("dW TRN2 Hp4096", 4096, 4096, 6144, True, True),
          ("dW TRN1 Hp4096", 4096, 4608, 6144, True, True), ("fwd TRN3 Hp4096", 6144, 1024, 4096, False, False),
          ("dX TRN3 Hp4096", 6144, 4096, 1024, False, True),
          ("square 4096", 4096, 4096, 4096, False, False), ("square 8192", 8192, 8192, 8192, False, False),
          ("fwd TRN1 Hp1024", 6144, 1024, 4608, False, False), ("fwd HxH", 6144, 1024, 1024, False, False)]
print(f"{'shape':18s} {'M':>5s} {'N':>5s} {'K':>5s} " + " ".join(f"{'v' + str(v) + ' us':>9s} {'TF/s':>6s}" for v in variants), flush=True)
for name, M, N, K, tA, tB in SHAPES:
    A = torch.randn((K, M) if tA else (M, K), device=dev).to(bf)
    B = torch.randn((K, N) if tB else (N, K), device=dev).to(bf)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if (tA and tB) else bf)
    cells = []
    for v in variants:
        lib.egk_gemm_set_pipeline(v)
        us = time_us(lambda: ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, out, N, transA=tA, transB=tB, accumulate=tA and tB,
                                      compute=ops.BF16, allow_splitk=False), 10)
        cells.append(f"{us:9.1f} {2.0 * M * N * K / us / 1e6:6.0f}")
    lib.egk_gemm_set_pipeline(1)
    print(f"{name:18s} {M:5d} {N:5d} {K:5d} " + " ".join(cells), flush=True)
