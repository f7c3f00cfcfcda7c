#!/usr/bin/env python3
"""The 192 x 128 tile with four loader waves (egk_gemm_set_pipeline(871)) against the same tile whose compute waves issue their own
LDS-DMA pieces (870): bit equality first (eager, one launch each), then device time per launch inside a hipGraph."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import _lib, ops

dev, dt = "cuda", torch.bfloat16
lib = _lib.load()


def time_us(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


lib.egk_gemm_set_pipeline(16)
for M in (6144, 192, 6000):
    for N, K, tB, name in [(1024, 1024, False, "fwd HxH"), (1024, 2048, False, "fwd K=2H"), (1024, 4608, False, "fwd TRN1"),
                           (1024, 1024, True, "dX HxH"), (2048, 1024, True, "dX N=2H"), (1024, 64, False, "K=64")]:
        torch.manual_seed(1)
        A = torch.randn(M, K, device=dev).to(dt)
        B = torch.randn((K, N) if tB else (N, K), device=dev).to(dt)
        bias = torch.randn(N, device=dev)
        outs = {}
        for knob in (870, 871):
            lib.egk_gemm_set_pipeline(knob)
            out = torch.zeros(M, N, device=dev, dtype=dt)
            ops.gemm(M, N, A, K, B, B.shape[1], K, out, N, transB=tB, bias=bias, act=1, compute=ops.BF16)
            torch.cuda.synchronize()
            outs[knob] = out
        same = torch.equal(outs[870], outs[871])
        cells = []
        if M == 6144:
            for knob in (870, 871):
                lib.egk_gemm_set_pipeline(knob)
                out = torch.empty(M, N, device=dev, dtype=dt)
                us = time_us(lambda: ops.gemm(M, N, A, K, B, B.shape[1], K, out, N, transB=tB, bias=bias, act=1, compute=ops.BF16))
                cells.append(f"{us:8.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF/s")
        print(f"{name + f' {M}x{N}x{K}':30s} bit-equal {same}   " + "   ".join(cells), flush=True)
lib.egk_gemm_set_pipeline(870)
lib.egk_gemm_set_pipeline(1)
