#!/usr/bin/env python3
"""The chain's contractions (6144-row outputs, row-major A) on 96 x 128 tiles (variant 8) against 192 x 128 tiles on 8 waves
(variant 16) and the policy's own choice (1): device time per launch inside a hipGraph.
python tools/round6/r192_bench.py [--iters 40] [--variants 8,16,1]"""
import argparse
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--variants", default="8,16,1")
ap.add_argument("--rows", default="6144,2048,8192")
args = ap.parse_args()
dev, dt = "cuda", torch.bfloat16


def time_us(fn, iters):
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


variants = [int(v) for v in args.variants.split(",")]
lib = _lib.load()
print(f"{'shape':34s} " + " ".join(f"{'v' + str(v) + ' us':>9s} {'TF/s':>6s}" for v in variants))
for M in [int(r) for r in args.rows.split(",")]:
    for N, K, tB, name in [(1024, 1024, False, "fwd HxH"), (1024, 2048, False, "fwd SAGE K=2H"), (1024, 4608, False, "fwd TRN1"),
                           (1024, 1024, True, "dX HxH"), (1024, 2048, True, "dX K=2H"), (4608, 1024, True, "dX TRN1 (N=4608)")]:
        A = torch.randn(M, K, device=dev).to(dt)
        B = torch.randn((K, N) if tB else (N, K), device=dev).to(dt)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt)

        def run():
            ops.gemm(M, N, A, K, B, B.shape[1], K, out, N, transB=tB, bias=bias, act=1, compute=ops.BF16)
        cells = []
        for v in variants:
            prev = lib.egk_gemm_set_pipeline(v)
            try:
                us = time_us(run, args.iters)
            finally:
                lib.egk_gemm_set_pipeline(prev)
            cells.append(f"{us:9.1f} {2.0 * M * N * K / us / 1e6:6.0f}")
        print(f"{name + f' {M}x{N}x{K}':34s} " + " ".join(cells), flush=True)
