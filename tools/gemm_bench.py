#!/usr/bin/env python3
"""Per-shape timing of the contractions of the bench workload (HIP events, back-to-back launches).
Usage: python tools/gemm_bench.py [--iters 50] [--dtype bf16|f32act]"""
import argparse
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--variants", default="1", help="comma list of egk_gemm_set_pipeline values (0 generic, 1 auto, 2/3/4)")
ap.add_argument("--layouts", action="store_true", help="same dims in the four operand layouts instead of the workload shapes")
ap.add_argument("--splitk", default="", help="comma list of forced split-K factors (applied to the dW shapes)")
ap.add_argument("--hp", type=int, default=0, help="the temporal pooling's contractions at this hidden size (e.g. 4096: BASELINE configs[1..2] as published) instead of the workload list")
args = ap.parse_args()
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
dev = "cuda"

# (name, M, N, K, transA, transB, out_f32/accumulate)
N6, H = 6144, 1024
SHAPES = [
    ("fwd TRN1 (per task)", 2048, 1024, 4608, False, False, False),
    ("fwd HxH merged", N6, H, H, False, False, False),
    ("fwd SAGE combine K=2H", N6, H, 2 * H, False, False, False),
    ("fwd head HxH", 2048, H, H, False, False, False),
    ("fwd cls 478", 2048, 478, H, False, False, True),
    ("dX HxH merged", N6, H, H, False, True, False),
    ("dX head", 2048, H, H, False, True, False),
    ("dX cls 478", 2048, H, 478, False, True, False),
    ("dW HxH merged", H, H, N6, True, True, True),
    ("dW TRN1 (per task)", H, 4608, 2048, True, True, True),
    ("dW TRN1 merged", H, 4608, N6, True, True, True),
    ("dW head", H, H, 2048, True, True, True),
    ("dW cls 478", 478, H, 2048, True, True, True),
    ("fwd cls 115", 2048, 115, H, False, False, True),
    ("dX cls 115 (K pad 128)", 2048, H, 128, False, True, False),
    ("dX cls 478 (K pad 512)", 2048, H, 512, False, True, False),
    ("dW cls 115", 115, H, 2048, True, True, True),
    ("dW cls 2", 2, H, 2048, True, True, True),
]
def time_us(fn, iters):
    """Device time per launch: ``iters`` launches captured in one hipGraph (no host launch cost in the number)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


if args.hp:
    HP = args.hp
    SHAPES = [("fwd TRN1 merged", N6, HP, 4608, False, False, False), ("fwd TRN2 merged", N6, HP, HP, False, False, False),
              ("fwd TRN3 merged", N6, H, HP, False, False, False),
              ("dX TRN3", N6, HP, H, False, True, False), ("dX TRN2", N6, HP, HP, False, True, False),
              ("dW TRN1 merged", HP, 4608, N6, True, True, True), ("dW TRN2 merged", HP, HP, N6, True, True, True),
              ("dW TRN3 merged", H, HP, N6, True, True, True)]
if args.layouts:
    SHAPES = []
    for (M, N, K) in [(2048, 2048, 2048), (4096, 2048, 2048), (4096, 4096, 4096), (8192, 8192, 8192)]:
        for nm, tA, tB in [("NN", False, False), ("NT", False, True), ("TT", True, True), ("TN", True, False)]:
            SHAPES.append((f"{nm} {M}x{N}x{K}", M, N, K, tA, tB, False))
variants = [int(v) for v in args.variants.split(",")]
from egopack_amd import _lib
print(f"{'shape':28s} {'M':>5s} {'N':>5s} {'K':>5s} splitk " + " ".join(f"{'v' + str(v) + ' us':>9s} {'TF/s':>6s}" for v in variants))
for name, M, N, K, tA, tB, f32out in SHAPES:
    pad8 = lambda n: (n + 7) // 8 * 8  # rows padded to 16 bytes, as ops._operand_rows builds them
    A = torch.randn((K, pad8(M)) if tA else (M, pad8(K)), device=dev).to(dt)
    B = torch.randn((K, pad8(N)) if tB else (N, pad8(K)), device=dev).to(dt)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if (f32out or dt == torch.float32) else dt)
    acc = tA and tB and not args.layouts
    sk = _lib.load().egk_gemm_splitk(M, N, K, ops.BF16)

    def run():
        ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, out, N, transA=tA, transB=tB, accumulate=acc, compute=ops.BF16)
    cells = []
    for v in variants:
        _lib.load().egk_gemm_set_pipeline(v)
        us = time_us(run, args.iters)
        cells.append(f"{us:9.1f} {2.0 * M * N * K / us / 1e6:6.0f}")
    _lib.load().egk_gemm_set_pipeline(1)
    print(f"{name:28s} {M:5d} {N:5d} {K:5d} {sk:6d} " + " ".join(cells))
    if args.splitk:
        row = []
        for fk in [int(v) for v in args.splitk.split(",")]:
            def run_sk():
                ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, out, N, transA=tA, transB=tB, accumulate=acc, compute=ops.BF16, splitk=fk)
            row.append(f"sk{fk}: {time_us(run_sk, args.iters):.1f}us")
        print("      forced split-K  " + "  ".join(row))
