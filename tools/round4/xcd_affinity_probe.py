#!/usr/bin/env python3
"""Does an XCD's L2 keep what a kernel wrote for the NEXT kernel?  Row LayerNorm chains on [6144, 1024] bf16:
  const   : every launch reads the same (never rewritten) tensor                       -- what tools/row_bench.py times
  chain   : launch i reads what launch i - 1 wrote, SAME row -> workgroup (-> XCD) map -- producer and consumer rows on one XCD
  gemm    : launch i reads what a contraction wrote (its tiles: 768-row patches per XCD) -- the situation inside the step
"""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
import torch
from _timing import time_us
from egopack_amd import ops

dev = "cuda"
N, H = 6144, 1024
x = torch.randn(N, H, device=dev).to(torch.bfloat16)
w, b = torch.randn(H, device=dev), torch.randn(H, device=dev)
W = (torch.randn(H, H, device=dev) * 0.03).to(torch.bfloat16)


def const():
    for _ in range(8):
        ops.row_layernorm(x, w, b, relu=True)


def chain():
    y = x
    for _ in range(8):
        y = ops.row_layernorm(y, w, b, relu=True)


def gemm_only():
    y = x
    for _ in range(8):
        y = ops.linear(y, W)


def gemm_ln():
    y = x
    for _ in range(8):
        y = ops.linear(y, W)
        y = ops.row_layernorm(y, w, b, relu=True)


with torch.no_grad(), ops.compute_mode("bf16"):
    t_const = time_us(const, 1) / 8
    t_chain = time_us(chain, 1) / 8
    t_gemm = time_us(gemm_only, 1) / 8
    t_pair = time_us(gemm_ln, 1) / 8
print(f"row LN reading a constant tensor {t_const:.1f} us; reading what the previous row LN wrote {t_chain:.1f} us")
print(f"contraction chain {t_gemm:.1f} us per launch; contraction + row LN pair {t_pair:.1f} us  => row LN behind a contraction {t_pair - t_gemm:.1f} us"
      f" (if the contraction kept its chained time)")
