#!/usr/bin/env python3
"""Feasibility of a SPLIT forward chain: the headline's forward pass is one chain of [row kernel -> contraction] pairs at M = 6144
rows (three tasks merged).  Would two / three independent chains of 3072 / 2048 rows on their own streams -- one chain's row
kernels hidden under the other's contractions -- finish sooner?  Captured graphs, replay time per variant."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch

from egopack_amd import ops

dev, bf = "cuda", torch.bfloat16
H, LAYERS, ROWS = 1024, 6, 6144
W = [torch.randn(H, H, device=dev).to(bf) * 0.03 for _ in range(LAYERS)]
Wb = [torch.randn(H, H, device=dev).to(bf) * 0.03 for _ in range(LAYERS)]
g_, b_ = torch.ones(H, device=dev), torch.zeros(H, device=dev)


def chain(x, wide):
    for l in range(LAYERS):
        y = ops.row_layernorm(x, g_, b_, 1e-5, relu=True)
        if wide and l % 2 == 0:  # a SAGE layer's two-source contraction (K = 2H)
            x = ops.linear(y, W[l], None, x2=x, W2=Wb[l])
        else:
            x = ops.linear(y, W[l], None)
    return x


def variant(n_streams, wide):
    xs = [torch.randn(ROWS // n_streams, H, device=dev).to(bf) for _ in range(n_streams)]
    side = [torch.cuda.Stream() for _ in range(n_streams - 1)]
    cap = torch.cuda.Stream()
    outs = []

    def body():
        main = torch.cuda.current_stream()
        for s in side:
            s.wait_stream(main)
        outs.clear()
        outs.append(chain(xs[0], wide))
        for s, x in zip(side, xs[1:]):
            with torch.cuda.stream(s):
                outs.append(chain(x, wide))
        for s in side:
            main.wait_stream(s)

    with torch.no_grad(), ops.compute_mode("bf16"):
        with torch.cuda.stream(cap):
            for _ in range(2):
                body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
            body()
    torch.cuda.synchronize()
    for _ in range(5):
        g.replay()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20 * 1e3)
    return best


for wide in (False, True):
    row = [f"{n} chain(s): {variant(n, wide):6.1f} us" for n in (1, 2, 3)]
    print(f"{LAYERS} x [row LayerNorm -> contraction{' (every other K = 2H)' if wide else ''}], {ROWS} rows in all: " + " | ".join(row), flush=True)
