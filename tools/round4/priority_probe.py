#!/usr/bin/env python3
"""Does a high-priority stream's launch get CU slots ahead of a long launch that already fills the chip on a low-priority stream?
L: a weight-gradient-shaped contraction (1024 x 4608 x 6144, ~120 us, every CU busy); H: a dX-shaped one (6144 x 1024 x 2048, ~43 us
alone) issued right behind it.  Reported: H's duration (HIP events on H) alone, beside L with equal priorities, with H high / L low."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

dev, BF = "cuda", torch.bfloat16
lo_p, hi_p = torch.cuda.Stream.priority_range()
print("priority range (least, greatest):", lo_p, hi_p)
dY = torch.randn(6144, 1024, device=dev).to(BF)
X = torch.randn(6144, 4608, device=dev).to(BF)
dW = torch.zeros(1024, 4608, device=dev)
A = torch.randn(6144, 2048, device=dev).to(BF)
W = torch.randn(2048, 1024, device=dev).to(BF)
out = torch.empty(6144, 1024, device=dev, dtype=BF)


def long_launch():
    ops.gemm(1024, 4608, dY, 1024, X, 4608, 6144, dW, 4608, transA=True, transB=True, accumulate=True, allow_splitk=False)


def short_launch():
    ops.gemm(6144, 1024, A, 2048, W, 1024, 2048, out, 1024, transB=True)


def run(ph, pl, beside=True, reps=20):
    H, L = torch.cuda.Stream(priority=ph), torch.cuda.Stream(priority=pl)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if beside:
            with torch.cuda.stream(L):
                long_launch()
        with torch.cuda.stream(H):
            e0.record()
            short_launch()
            e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for _ in range(3):
    long_launch(); short_launch()
torch.cuda.synchronize()
print(f"H alone: {run(0, 0, beside=False):.1f} us")
print(f"H beside L, equal priorities: {run(0, 0):.1f} us")
print(f"H high ({hi_p}) beside L low ({lo_p}): {run(hi_p, lo_p):.1f} us")
print(f"H low beside L high: {run(lo_p, hi_p):.1f} us")
