#!/usr/bin/env python3
"""Grouped row-major launches of the task heads (3 x [2048, 1024] x [1024, 1024], NN forward / NT dX) under every tile
variant the grouped entry point takes, against one merged 6144-row launch of the same work."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
H, M, G = 1024, 2048, 3
bf = torch.bfloat16
x = torch.randn(G * M, H, device="cuda").to(bf)
Ws = [(torch.randn(H, H, device="cuda") * 0.03).to(bf) for _ in range(G)]
y = torch.empty(G * M, H, device="cuda", dtype=bf)
for tB, name in ((False, "NN fwd"), (True, "NT dX ")):
    probs = [((M, H, x[g * M:(g + 1) * M], H, Ws[g], H, H, y[g * M:(g + 1) * M], H), dict(transB=tB, compute=ops.BF16)) for g in range(G)]
    fl = G * 2.0 * M * H * H
    for knob, kn in ((1, "policy"), (2, "128 3-stage"), (3, "128 2-stage"), (4, "128 4-stage"), (8, "96-row"), (11, "64-row"), (5, "two wave groups")):
        lib.egk_gemm_set_pipeline(knob)
        try:
            us = time_us(lambda: ops.gemm_grouped(probs), 20)
            print(f"{name} grouped x{G} [{kn:16s}] {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
        except Exception as e:  # noqa: BLE001
            print(f"{name} grouped x{G} [{kn:16s}] not available: {str(e)[:80]}")
    lib.egk_gemm_set_pipeline(1)
    us = time_us(lambda: ops.gemm(G * M, H, x, H, Ws[0], H, H, y, H, transB=tB, compute=ops.BF16), 20)
    print(f"{name} ONE merged {G * M}-row launch      {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
    us = time_us(lambda: [ops.gemm(*a, **k) for a, k in probs], 20)
    print(f"{name} {G} separate launches            {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
