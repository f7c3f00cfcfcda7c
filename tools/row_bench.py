#!/usr/bin/env python3
"""Timing of the HBM-bound row kernels at the bench shape [6144, 1024] (device time inside a hipGraph)."""
import sys

sys.path.insert(0, ".")
import torch

from egopack_amd import data as D
from egopack_amd import ops

dev = "cuda"
N, H = 6144, 1024
dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
if len(sys.argv) > 3:  # row_bench.py bf16 <cap_partial> <cap_wide>
    from egopack_amd import _lib
    _lib.load().egk_tune(1, int(sys.argv[2]))
    _lib.load().egk_tune(2, int(sys.argv[3]))
    print("caps", sys.argv[2], sys.argv[3])
import os
if os.environ.get("EGK_ROWS_V2") is not None:  # EGK_ROWS_V2=0: the generic row kernels (egk_tune 3)
    from egopack_amd import _lib
    _lib.load().egk_tune(3, int(os.environ["EGK_ROWS_V2"]))
    print("rows v2", os.environ["EGK_ROWS_V2"])
x = torch.randn(N, H, device=dev).to(dt)
g = torch.randn(N, H, device=dev).to(dt)
w, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
seg = torch.tensor([0, 2048, 4096, 6144], dtype=torch.int32, device=dev)
ei = torch.cat([D.radius_band_edges(torch.arange(32), 1) + 32 * i for i in range(N // 32)], 1)
graph = D.build_csr(ei, N).to(dev)
pos = torch.arange(N, device=dev) % 32
freq = torch.logspace(0, 1, H // 2, 1e-4).to(dev)


sys.path.insert(0, "tools")
from _timing import time_us


def timeit(name, fn, bytes_, iters=20):
    us = time_us(fn, iters)
    print(f"{name:28s} {us:8.1f} us  {bytes_ / us / 1e3:8.0f} GB/s", flush=True)


eb = x.element_size()
nb = N * H * eb
timeit("csr_mean_aggregate fwd", lambda: ops.csr_mean_aggregate(x, graph), 2 * nb)
timeit("row_layernorm relu fwd", lambda: ops.row_layernorm(x, w, b, relu=True), 2 * nb)
timeit("row_layernorm relu+drop fwd", lambda: ops.row_layernorm(x, w, b, relu=True, p=0.5, training=True), 2 * nb + N * H)
timeit("graph_layernorm_lrelu fwd", lambda: ops.graph_layernorm_lrelu(x, w, b, seg), 3 * nb)
timeit("pe_add", lambda: ops.pe_add(x, pos, freq), 2 * nb)
xr = x.clone().requires_grad_(True)
wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)


def fb(f, ins):
    def run():
        y = f()
        torch.autograd.grad(y, ins, g)
    return run


# forward + backward inside one capture (a backward alone would run on the forward's, uncaptured, stream)
timeit("row_layernorm fwd+bwd", fb(lambda: ops.row_layernorm(xr, wr, br, relu=True), (xr, wr, br)), 5 * nb)
timeit("graph_layernorm fwd+bwd", fb(lambda: ops.graph_layernorm_lrelu(xr, wr, br, seg), (xr, wr, br)), 8 * nb)
timeit("csr_mean_aggregate fwd+bwd", fb(lambda: ops.csr_mean_aggregate(xr, graph), (xr,)), 4 * nb)
out = torch.empty(H, device=dev)
timeit("colsum", lambda: ops._colsum_into(g, out, False), nb)
# floor: what a plain device copy / elementwise kernel of the same tensors costs here
y_ = torch.empty_like(x)
timeit("torch copy_ (floor)", lambda: y_.copy_(x), 2 * nb)
timeit("torch add (2 reads 1 write)", lambda: torch.add(x, g, out=y_), 3 * nb)
big = torch.randn(64 * N, H, device=dev).to(dt)
big2 = torch.empty_like(big)
timeit("torch copy_ 64x larger", lambda: big2.copy_(big), 2 * 64 * nb, iters=5)
