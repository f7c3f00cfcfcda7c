#!/usr/bin/env python3
"""Grouped weight-gradient launch (4 x [1024 x 1024], K = nodes) against the same problems as separate launches."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
H = 1024
for M in (2048, 6144, 16384):
    gs = [torch.randn(M, H, device="cuda").to(torch.bfloat16) for _ in range(4)]
    xs = [torch.randn(M, H, device="cuda").to(torch.bfloat16) for _ in range(4)]
    outs = [torch.zeros(H, H, device="cuda") for _ in range(4)]
    db = [torch.zeros(H, device="cuda") for _ in range(4)]
    probs = [((H, H, gs[i], H, xs[i], H, M, outs[i], H), dict(transA=True, transB=True, accumulate=True, compute=ops.BF16, dbias=db[i]))
             for i in range(4)]

    def separate():
        for a, kw in probs:
            ops.gemm(*a, **kw)
    fl = 4 * 2.0 * H * H * M
    for knob, name in ((1, "policy"), (3, "4-wave 128x128"), (5, "two wave groups")):
        lib.egk_gemm_set_pipeline(knob)
        us = time_us(lambda: ops.gemm_grouped(probs), 20)
        print(f"M={M:6d} grouped x4 [{name:16s}] {us:7.1f} us  {fl / us / 1e6:7.0f} TF/s")
    lib.egk_gemm_set_pipeline(1)
    us = time_us(separate, 20)
    print(f"M={M:6d} 4 separate (split-K policy)   {us:7.1f} us  {fl / us / 1e6:7.0f} TF/s")
    for n in (2, 3):
        us = time_us(lambda: ops.gemm_grouped(probs[:n]), 20)
        print(f"M={M:6d} grouped x{n}                     {us:7.1f} us  {n / 4 * fl / us / 1e6:7.0f} TF/s")
    gs8 = gs + [torch.randn(M, H, device="cuda").to(torch.bfloat16) for _ in range(4)]
    xs8 = xs + [torch.randn(M, H, device="cuda").to(torch.bfloat16) for _ in range(4)]
    outs8 = outs + [torch.zeros(H, H, device="cuda") for _ in range(4)]
    probs8 = [((H, H, gs8[i], H, xs8[i], H, M, outs8[i], H), dict(transA=True, transB=True, accumulate=True, compute=ops.BF16))
              for i in range(8)]
    for knob, name in ((1, "policy"), (3, "4-wave 128x128"), (5, "two wave groups")):
        lib.egk_gemm_set_pipeline(knob)
        for n in (6, 8):
            us = time_us(lambda: ops.gemm_grouped(probs8[:n]), 10)
            print(f"M={M:6d} grouped x{n} [{name:16s}] {us:7.1f} us  {n / 4 * fl / us / 1e6:7.0f} TF/s")
    lib.egk_gemm_set_pipeline(1)
