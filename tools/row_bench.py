#!/usr/bin/env python3
"""Timing of the HBM-bound row kernels at the bench shape [6144, 1024] (HIP events, back-to-back launches)."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import data as D
from egopack_amd import ops

dev = "cuda"
N, H = 6144, 1024
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
x = torch.randn(N, H, device=dev).to(dt)
g = torch.randn(N, H, device=dev).to(dt)
w, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
seg = torch.tensor([0, 2048, 4096, 6144], dtype=torch.int32, device=dev)
ei = torch.cat([D.radius_band_edges(torch.arange(32), 1) + 32 * i for i in range(N // 32)], 1)
graph = D.build_csr(ei, N).to(dev)
pos = torch.arange(N, device=dev) % 32
freq = torch.logspace(0, 1, H // 2, 1e-4).to(dev)


def timeit(name, fn, bytes_, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:28s} {us:8.1f} us  {bytes_ / us / 1e3:8.0f} GB/s")


eb = x.element_size()
nb = N * H * eb
timeit("csr_mean_aggregate fwd", lambda: ops.csr_mean_aggregate(x, graph), 2 * nb)
timeit("row_layernorm relu fwd", lambda: ops.row_layernorm(x, w, b, relu=True), 2 * nb)
timeit("row_layernorm relu+drop fwd", lambda: ops.row_layernorm(x, w, b, relu=True, p=0.5, training=True), 2 * nb + N * H)
timeit("graph_layernorm_lrelu fwd", lambda: ops.graph_layernorm_lrelu(x, w, b, seg), 3 * nb)
timeit("pe_add", lambda: ops.pe_add(x, pos, freq), 2 * nb)
xr = x.clone().requires_grad_(True)
wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
y = ops.row_layernorm(xr, wr, br, relu=True)
timeit("row_layernorm bwd", lambda: torch.autograd.grad(y, (xr, wr, br), g, retain_graph=True), 3 * nb)
y2 = ops.graph_layernorm_lrelu(xr, wr, br, seg)
timeit("graph_layernorm bwd", lambda: torch.autograd.grad(y2, (xr, wr, br), g, retain_graph=True), 5 * nb)
y3 = ops.csr_mean_aggregate(xr, graph)
timeit("csr_mean_aggregate bwd", lambda: torch.autograd.grad(y3, (xr,), g, retain_graph=True), 2 * nb)
out = torch.empty(H, device=dev)
timeit("colsum", lambda: ops._colsum_into(g, out, False), nb)
