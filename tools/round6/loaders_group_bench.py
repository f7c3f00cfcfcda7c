#!/usr/bin/env python3
"""Weight-gradient groups (A^T B, fused bias gradient) on 128 x 128 tiles: compute waves that issue their own LDS-DMA pieces (880)
against four loader waves on a 2-stage (881) / 3-stage (882) ring: bit equality, then device time per launch inside a hipGraph."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import _lib, ops

dev = "cuda"
lib = _lib.load()


def time_us(fn, iters=20):
    for _ in range(2):
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


for name, sizes in [("8 x HxH, 6144 rows", [(1024, 1024, 6144)] * 8), ("6 x HxH, 6144 rows", [(1024, 1024, 6144)] * 6),
                    ("heads-like", [(640, 1024, 64), (640, 1024, 2048), (1024, 1024, 64), (1024, 1024, 2048)] * 2),
                    ("8 x HxH, 16384 rows", [(1024, 1024, 16384)] * 8), ("pooling tail, 6144 rows", [(1024, 1024, 6144), (1024, 1024, 6144), (1024, 4608, 6144)]), ("ragged", [(1000, 520, 2048), (320, 192, 512), (1024, 1024, 1024)])]:
    g = torch.Generator().manual_seed(5)
    ops_in = []
    for (M, N, K) in sizes:
        A = torch.randn(K, M, generator=g).to(torch.bfloat16).to(dev)
        B = torch.randn(K, N, generator=g).to(torch.bfloat16).to(dev)
        ops_in.append((A, B, torch.randn(M, N, generator=g).to(dev), torch.randn(M, generator=g).to(dev)))
    outs, times = {}, {}
    for knob, pre in ((880, 1), (881, 1), (882, 1), (1880, 13), (1883, 13)):  # (1xxx: the 256 x 128 tile forced; 883: + loader waves)
        lib.egk_gemm_set_pipeline(pre)
        lib.egk_gemm_set_pipeline(knob % 1000)
        cs = [(c.clone(), b.clone()) for _, _, c, b in ops_in]
        probs = [((M, N, A, A.stride(0), B, B.stride(0), K, c, N), dict(transA=True, transB=True, compute=ops.BF16, accumulate=True, dbias=b))
                 for (M, N, K), (A, B, _, _), (c, b) in zip(sizes, ops_in, cs)]
        ops.gemm_grouped(probs)
        torch.cuda.synchronize()
        outs[knob] = cs
        if "rows" in name:
            times[knob] = time_us(lambda: ops.gemm_grouped(probs))
    same = all(all(torch.equal(c0, o[i][0]) and torch.equal(b0, o[i][1]) for o in outs.values()) for i, (c0, b0) in enumerate(outs[880]))
    if not same:
        for k, o in outs.items():
            bad = [(i, float((o[i][0] - c0).abs().max()), float((o[i][1] - b0).abs().max())) for i, (c0, b0) in enumerate(outs[880])
                   if not (torch.equal(c0, o[i][0]) and torch.equal(b0, o[i][1]))]
            if bad:
                print(f"   knob {k}: problems (index, max |dC|, max |dbias|) {bad}")
    print(f"{name:24s} bit-equal {same}  " + "  ".join(f"{k}: {v:7.1f} us" for k, v in times.items()), flush=True)
lib.egk_gemm_set_pipeline(880)
lib.egk_gemm_set_pipeline(1)
