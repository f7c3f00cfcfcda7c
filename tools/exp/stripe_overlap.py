#!/usr/bin/env python3
"""One chain of L (contraction + row kernel) pairs over 6144 rows against the same rows cut into S stripes, every stripe
a chain of its own on its own stream (captured hipGraph, replay time).  Usage: python tools/exp/stripe_overlap.py"""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

dev, bf = "cuda", torch.bfloat16
MT, N, L = 6144, 1024, 12


def timed(g, n=20):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, rows in ((1024, False), (2048, False), (1024, True)):
    W = torch.randn(N, K, device=dev).to(bf) * 0.03
    X = torch.randn(MT, K, device=dev).to(bf)
    Y = torch.empty(MT, N, device=dev, dtype=bf)
    Z = torch.empty(MT, N, device=dev, dtype=bf)
    lw, lb = torch.ones(N, device=dev), torch.zeros(N, device=dev)

    def chain(r0, r1):
        m = r1 - r0
        for _ in range(L):
            ops.gemm(m, N, X[r0:r1], K, W, K, K, Y[r0:r1], N, compute=ops.BF16)
            if rows:
                ops.row_layernorm(Y[r0:r1], lw, lb, relu=True)

    side = [torch.cuda.Stream() for _ in range(6)]
    cap = torch.cuda.Stream()
    chain(0, MT)
    torch.cuda.synchronize()
    res = []
    for S in (1, 2, 3, 4, 6):
        b = [MT * i // S // 96 * 96 for i in range(S)] + [MT]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap), torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            if S == 1:
                chain(0, MT)
            else:
                ev = torch.cuda.Event()
                ev.record(cap)
                for i in range(S):
                    side[i].wait_event(ev)
                    with torch.cuda.stream(side[i]):
                        chain(b[i], b[i + 1])
                for i in range(S):
                    cap.wait_stream(side[i])
        res.append((S, timed(g) / L))
    print(f"K={K} row kernel={rows}: " + "  ".join(f"S={s}: {t:6.1f} us/pair" for s, t in res))
