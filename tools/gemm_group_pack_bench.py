#!/usr/bin/env python3
"""Grouped launches: XCD-packed placement (an XCD works through a consecutive run of the launch's tile list: one problem at a
time) against the spread placement (every problem's tiles dealt over all eight XCDs), same kernels, same results.
Weight-gradient groups of the headline step (x6, x4, the tail group with the 4608-wide TRN weight) and the heads' forward groups."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops

lib = _lib.load()
H, M = 1024, 6144
bf = torch.bfloat16


def dw(n_out, n_in, rows):
    g, x = torch.randn(rows, n_out, device="cuda").to(bf), torch.randn(rows, n_in, device="cuda").to(bf)
    out = torch.zeros(n_out, n_in, device="cuda")
    return ((n_out, n_in, g, n_out, x, n_in, rows, out, n_in), dict(transA=True, transB=True, accumulate=True, compute=ops.BF16))


def fwd(rows, n_out, k):
    x, w = torch.randn(rows, k, device="cuda").to(bf), torch.randn(n_out, k, device="cuda").to(bf)
    out = torch.empty(rows, n_out, device="cuda", dtype=bf)
    return ((rows, n_out, x, k, w, k, k, out, n_out), dict(compute=ops.BF16))


cases = {"dW x6 HxH (K = 6144)": [dw(H, H, M) for _ in range(6)], "dW x4 HxH": [dw(H, H, M) for _ in range(4)],
         "dW x8 HxH (K = 2048: heads)": [dw(H, H, 2048) for _ in range(8)],
         "dW tail: 4608-wide + 2 HxH": [dw(H, H, M), dw(H, H, M), dw(H, 4608, M)],
         "fwd x3 heads 2048 x 1024 x 1024": [fwd(2048, H, H) for _ in range(3)]}
for name, probs in cases.items():
    fl = sum(2.0 * a[0] * a[1] * a[6] for a, _ in probs)
    row = []
    for knob, tag in ((300, "spread"), (301, "packed")):
        lib.egk_gemm_set_pipeline(knob)
        for four in (False, True):
            if four and len(probs) > 4:
                continue
            us = min(time_us(lambda: ops.gemm_grouped(probs, four_wave=four), 20) for _ in range(3))
            row.append(f"{tag}{'/4w' if four else ''} {us:6.1f} us ({fl / us / 1e6:5.0f} TF/s)")
    lib.egk_gemm_set_pipeline(301)
    print(f"{name:34s} " + "   ".join(row))
