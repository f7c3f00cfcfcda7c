#!/usr/bin/env python3
"""The step's last weight gradient (first TRN linear: dW [1024, 4608] over K = 6144 nodes, the k-major 'tt' form, f32 accumulate
into the gradient slot) on 128 x 128 tiles (policy), on 256 x 256 tiles with K split over 1-4 slabs (egk_gemm_set_pipeline(7),
splitk), and the TRN trio as the step launches it (one grouped launch)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
dev, bf = "cuda", torch.bfloat16
for (M, N, K) in [(1024, 4608, 6144), (1024, 1024, 6144), (1024, 4608, 2048)]:
    A = torch.randn(K, M, device=dev).to(bf)
    B = torch.randn(K, N, device=dev).to(bf)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32)
    row = []
    for v, sk in [(1, None), (3, 1), (7, 1), (7, 2), (7, 3), (7, 4), (3, 2)]:
        lib.egk_gemm_set_pipeline(v)
        try:
            us = time_us(lambda: ops.gemm(M, N, A, M, B, N, K, out, N, transA=True, transB=True, accumulate=True, compute=ops.BF16,
                                          allow_splitk=sk is None, splitk=sk), 10)
            row.append(f"v{v}/sk{sk}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF/s")
        except Exception as e:  # noqa: BLE001
            row.append(f"v{v}/sk{sk}: {type(e).__name__}")
    lib.egk_gemm_set_pipeline(1)
    print(f"dW {M}x{N} K={K}: " + " | ".join(row), flush=True)
# the trio as one grouped launch
K = 6144
ops_ = []
for (M, N) in [(1024, 1024), (1024, 1024), (1024, 4608)]:
    A = torch.randn(K, M, device=dev).to(bf)
    B = torch.randn(K, N, device=dev).to(bf)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32)
    ops_.append(((M, N, A, M, B, N, K, out, N), dict(transA=True, transB=True, accumulate=True, compute=ops.BF16)))
print(f"trio grouped: {time_us(lambda: ops.gemm_grouped(ops_), 10):7.1f} us; the two small ones grouped: "
      f"{time_us(lambda: ops.gemm_grouped(ops_[:2]), 10):7.1f} us", flush=True)
