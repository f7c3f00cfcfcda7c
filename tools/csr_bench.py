#!/usr/bin/env python3
"""csr_gather forward / transposed-gated backward on the graphs of the bench workloads (device time in a hipGraph)."""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, data as D, ops

H = 1024
lib = _lib.load()
for name, T, B, lta in [("band T=32 B=192", 32, 192, False), ("band T=256 B=48", 256, 48, False), ("LTA T=32 B=64", 32, 64, True),
                        ("LTA T=256 B=16", 256, 16, True), ("mtl4 T=256 (3 band + 1 LTA) x16", 256, 64, "mix")]:
    eis, off = [], 0
    for b in range(B):
        if lta is True or (lta == "mix" and b % 4 == 3):
            y = torch.zeros(T, 2, dtype=torch.long) + 1
            y[:2] = -1
            ei = D.lta_connectivity_edges(torch.arange(T), y, 1.5)
        else:
            ei = D.radius_band_edges(torch.arange(T), 1)
        eis.append(ei + off)
        off += T
    N = off
    g = D.build_csr(torch.cat(eis, 1), N).to("cuda")
    x = torch.randn(N, H, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(x)

    def fwd():
        ops._csr_gather(x, g.rowptr, g.col, None, None, out, g.heavy)

    def bwd():
        ops._csr_gather(x, g.t_rowptr, g.t_col, g.t_wgt, x, out, g.t_heavy, g.t_heavy_mode)
    def bwd_inkernel():  # heavy_mode 1: every listed row summed by one workgroup of the same launch
        ops._csr_gather(x, g.t_rowptr, g.t_col, g.t_wgt, x, out, g.t_heavy, 1)

    def bwd_split():  # heavy_mode 0: two extra launches
        ops._csr_gather(x, g.t_rowptr, g.t_col, g.t_wgt, x, out, g.t_heavy, 0)
    deg = (g.t_rowptr[1:] - g.t_rowptr[:-1]).max().item()
    print(f"{name:34s} N={N:6d} E={g.col.numel():7d} max out-degree {deg:4d} ({g.t_heavy.numel()} listed)   fwd {time_us(fwd, 20):7.1f} us   "
          f"bwd (mode {g.t_heavy_mode}) {time_us(bwd, 20):7.1f} us   split launches {time_us(bwd_split, 20):7.1f} us   in-launch blocks {time_us(bwd_inkernel, 20):7.1f} us")
