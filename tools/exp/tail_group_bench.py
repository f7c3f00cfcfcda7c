#!/usr/bin/env python3
"""The step's tail as ONE grouped launch: dW of the three TRN linears ([1024 x 1024] x 2 and [1024 x 4608], K = 6144 nodes)
against the first-layer weight gradient alone + the two small ones as a group of their own."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import ops

bf, M = torch.bfloat16, 6144
g1 = torch.randn(M, 1024, device="cuda").to(bf)
x1 = torch.randn(M, 4608, device="cuda").to(bf)
gs = [torch.randn(M, 1024, device="cuda").to(bf) for _ in range(2)]
xs = [torch.randn(M, 1024, device="cuda").to(bf) for _ in range(2)]
dW1 = torch.zeros(1024, 4608, device="cuda")
dWs = [torch.zeros(1024, 1024, device="cuda") for _ in range(2)]
db = [torch.zeros(1024, device="cuda") for _ in range(3)]
kw = lambda i: dict(transA=True, transB=True, accumulate=True, compute=ops.BF16, dbias=db[i])
p1 = ((1024, 4608, g1, 1024, x1, 4608, M, dW1, 4608), kw(0))
ps = [((1024, 1024, gs[i], 1024, xs[i], 1024, M, dWs[i], 1024), kw(1 + i)) for i in range(2)]
fl1, fls = 2.0 * 1024 * 4608 * M, 2 * 2.0 * 1024 * 1024 * M
t1 = time_us(lambda: ops.gemm(*p1[0], **p1[1]), 20)
t2 = time_us(lambda: ops.gemm_grouped(ps), 20)
t3 = time_us(lambda: ops.gemm_grouped([p1] + ps), 20)
t4 = time_us(lambda: ops.gemm_grouped(ps + [p1]), 20)
print(f"first-layer dW alone            {t1:7.1f} us {fl1 / t1 / 1e6:6.0f} TF/s")
print(f"two H x H dW as a group         {t2:7.1f} us {fls / t2 / 1e6:6.0f} TF/s")
print(f"all three in ONE launch (big first) {t3:7.1f} us {(fl1 + fls) / t3 / 1e6:6.0f} TF/s   (sum of the two above {t1 + t2:.1f})")
print(f"all three in ONE launch (big last)  {t4:7.1f} us {(fl1 + fls) / t4 / 1e6:6.0f} TF/s")
