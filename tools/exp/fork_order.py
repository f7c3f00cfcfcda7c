#!/usr/bin/env python3
"""Does the ORDER in which a fork's two children are captured decide which of them stays on the parent's hardware queue?
A chain of L contractions on the capture stream; after every link a side-stream contraction is forked (weight-gradient
shaped).  (a) side launch captured BEFORE the chain's next link (what a backward pass does naturally), (b) captured AFTER
it, behind an event recorded at the fork point.  Same DAG, same work; prints the replay time of both."""
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


def link(i):
    ops.gemm(M, H, ys[i & 1] if i else x, H, W, H, H, ys[(i + 1) & 1], H, transB=True, compute=ops.BF16)


def side_work(i):
    ops.gemm(H, H, g, H, x, H, M, dWs[i], H, transA=True, transB=True, accumulate=True, compute=ops.BF16)


def timed(gr, n=20):
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


cap, side = torch.cuda.Stream(), torch.cuda.Stream()
for i in range(2):
    link(i)
    side_work(i)
torch.cuda.synchronize()
res = {}
for mode in ("side first", "chain first", "no side work"):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap), torch.cuda.graph(gr, stream=cap, capture_error_mode="thread_local"):
        pending = None
        for i in range(L):
            link(i)
            if mode == "no side work":
                continue
            if pending is not None:  # (chain first: the previous fork's side launch is captured now, after this link)
                ev, j = pending
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    side_work(j)
                pending = None
            if mode == "side first":
                side.wait_stream(cap)
                with torch.cuda.stream(side):
                    side_work(i)
            else:
                ev = torch.cuda.Event()
                ev.record(cap)
                pending = (ev, i)
        if pending is not None:
            ev, j = pending
            side.wait_event(ev)
            with torch.cuda.stream(side):
                side_work(j)
        cap.wait_stream(side)
    res[mode] = timed(gr)
    print(f"{mode:13s}: {res[mode]:8.1f} us per replay ({L} links)")
