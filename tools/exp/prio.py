#!/usr/bin/env python3
"""Do stream priorities let a dX-like chain keep its pace beside weight-gradient-like launches?  Eager launches (priorities
are a property of the stream; what a captured graph keeps of them is a separate question): a chain of L contractions on
stream A, one forked weight-gradient launch per link on stream B; A / B at normal / normal, high / low priority."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

dev, bf = "cuda", torch.bfloat16
M, H, L = 6144, 1024, 12
x = torch.randn(M, H, device=dev).to(bf)
g = torch.randn(M, H, device=dev).to(bf)
W = (torch.randn(H, H, device=dev) * 0.03).to(bf)
ys = [torch.empty(M, H, device=dev, dtype=bf) for _ in range(2)]
dWs = [torch.zeros(H, H, device=dev) for _ in range(L)]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)


def run(a, b, n=10):
    def once():
        ev = None
        for i in range(L):
            with torch.cuda.stream(a):
                ops.gemm(M, H, ys[i & 1] if i else x, H, W, H, H, ys[(i + 1) & 1], H, transB=True, compute=ops.BF16)
                ev = torch.cuda.Event()
                ev.record(a)
            b.wait_event(ev)
            with torch.cuda.stream(b):
                ops.gemm(H, H, g, H, x, H, M, dWs[i], H, transA=True, transB=True, accumulate=True, compute=ops.BF16)
        a.wait_stream(b)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(a)
    for _ in range(n):
        once()
    e1.record(a)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, pa, pb in (("normal / normal", 0, 0), ("high / normal", hi, 0), ("high / low", hi, lo), ("normal / low", 0, lo)):
    a, b = torch.cuda.Stream(priority=pa), torch.cuda.Stream(priority=pb)
    print(f"chain {name:16s}: {run(a, b):8.1f} us per pass")
