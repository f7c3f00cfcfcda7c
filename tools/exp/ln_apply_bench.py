#!/usr/bin/env python3
"""graph-LN apply passes alone: how much of a launch is the per-workgroup re-reduction of the statistics partials?"""
import sys

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us

from egopack_amd import _lib, ops
from egopack_amd.ops import _ck, _dt, _p, _stream

lib = _lib.load()
N, H = 6144, 1024
x = torch.randn(N, H, device="cuda").to(torch.bfloat16)
dy = torch.randn(N, H, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
dx = torch.empty_like(x)
w, b = torch.randn(H, device="cuda"), torch.randn(H, device="cuda")
seg = torch.tensor([0, 2048, 4096, 6144], dtype=torch.int32, device="cuda")
stats = torch.empty(6, device="cuda")
for nb in (512, 64, 1):
    part = torch.rand(nb * 3 * 2, dtype=torch.float64, device="cuda") * 1e3 + 1e6
    part[1::2] *= 10
    us = time_us(lambda: _ck(lib.egk_graphln_fwd_apply(_stream(), _p(x), _p(w), _p(b), _p(y), _p(stats), _p(seg), 3, N, H, 1e-5, 0.2,
                                                      _p(part), nb, _dt(x)), "fwd"), 20)
    ws_col = torch.empty(lib.egk_rowln_bwd_ws_rows(N) * 2 * H, device="cuda")
    us2 = time_us(lambda: _ck(lib.egk_graphln_bwd_apply(_stream(), _p(dy), _p(x), _p(w), _p(b), _p(stats), _p(dx), _p(seg), 3, N, H,
                                                        1e-5, 0.2, _p(part), nb, _p(ws_col), _dt(x)), "bwd"), 20)
    print(f"partials {nb:4d}: fwd apply {us:6.2f} us   bwd apply {us2:6.2f} us")
