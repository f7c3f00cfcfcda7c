#!/usr/bin/env python3
"""Do parallel branches of a captured hipGraph overlap on replay?  (a) three chains of 20 small contractions forked /
joined with stream events, (b) the same chains as the BACKWARD of three autograd branches run on three streams.
Prints replay time against the one-stream capture of the same work.  Usage: python tools/exp/branch_overlap.py"""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import ops

dev = "cuda"
bf = torch.bfloat16
M, N, K, L = 2048, 1024, 1024, 20
W = [torch.randn(N, K, device=dev).to(bf) for _ in range(3)]
X = [torch.randn(M, K, device=dev).to(bf) for _ in range(3)]
Y = [torch.empty(M, N, device=dev, dtype=bf) for _ in range(3)]


def chain(i):
    for _ in range(L):
        ops.gemm(M, N, X[i], K, W[i], K, K, Y[i], N, compute=ops.BF16)


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


side = [torch.cuda.Stream() for _ in range(3)]
cap = torch.cuda.Stream()
for i in range(3):
    chain(i)
torch.cuda.synchronize()

g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap), torch.cuda.graph(g1, stream=cap, capture_error_mode="thread_local"):
    for i in range(3):
        chain(i)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap), torch.cuda.graph(g3, stream=cap, capture_error_mode="thread_local"):
    ev = torch.cuda.Event()
    ev.record(cap)
    for i in range(3):
        side[i].wait_event(ev)
        with torch.cuda.stream(side[i]):
            chain(i)
    for i in range(3):
        cap.wait_stream(side[i])
print(f"(a) stream API: one stream {timed(g1):8.1f} us   three forked streams {timed(g3):8.1f} us")


class Chain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, i):
        ctx.i = i
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        chain(ctx.i)
        return g, None


def fwd_bwd(parallel):
    x = torch.ones(8, device=dev, requires_grad=True)
    outs = []
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for i in range(3):
        if parallel:
            side[i].wait_event(ev)
            with torch.cuda.stream(side[i]):
                outs.append(Chain.apply(x * 1.0, i).sum())
        else:
            outs.append(Chain.apply(x * 1.0, i).sum())
    if parallel:
        for i in range(3):
            main.wait_stream(side[i])
    torch.stack(outs).sum().backward()


for par in (False, True):
    with torch.cuda.stream(cap):
        fwd_bwd(par)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap), torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
        fwd_bwd(par)
    print(f"(b) autograd backward, branches on {'three streams' if par else 'one stream   '}: {timed(g):8.1f} us")


def fwd_bwd_manual():
    """(c) one backward call per branch, each issued inside its stream context"""
    main = torch.cuda.current_stream()
    x = torch.ones(8, device=dev)
    ev = torch.cuda.Event()
    ev.record(main)
    for i in range(3):
        side[i].wait_event(ev)
        with torch.cuda.stream(side[i]):
            leaf = x.clone().requires_grad_(True)
            out = Chain.apply(leaf * 1.0, i).sum()
            out.backward()
    for i in range(3):
        main.wait_stream(side[i])


with torch.cuda.stream(cap):
    fwd_bwd_manual()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap), torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
    fwd_bwd_manual()
print(f"(c) one backward call per branch inside its stream context: {timed(g):8.1f} us")
