#!/usr/bin/env python3
"""row LayerNorm + ReLU + dropout, forward and backward, bf16, against the oracle's storage model on the same keep mask."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from egopack_amd import ops  # noqa: E402
from oracle import storage as S  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))


for p in (0.0, 0.5):
    for rows in (512, 6144):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(rows, 1024, device="cuda", generator=g).bfloat16().requires_grad_(True)
        w = (1 + 0.1 * torch.randn(1024, device="cuda", generator=g)).requires_grad_(True)
        b = (0.1 * torch.randn(1024, device="cuda", generator=g)).requires_grad_(True)
        R = torch.randn(rows, 1024, device="cuda", generator=g).bfloat16()
        with ops.compute_mode("bf16"), ops.tap_dropout_masks() as masks:
            y = ops.row_layernorm(x, w, b, relu=True, p=p, training=True)
            y.backward(R)
            torch.cuda.synchronize()
        m = masks[0].cpu() if masks else None
        xc = x.detach().float().cpu().requires_grad_(True)
        wc, bc = w.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
        with S.bf16_storage():
            h = F.relu(F.layer_norm(xc, (1024,), wc, bc, 1e-5))
            if m is not None:
                h = h * m.float() / (1 - p)
            o = S.act(h)
            (o * R.float().cpu()).sum().backward()
        print(f"p={p} rows={rows}: y {rel(y.float().cpu(), o.detach()):.2e}  dx {rel(x.grad.float().cpu(), S._r(xc.grad)):.2e}  "
              f"dw {rel(w.grad.cpu(), wc.grad):.2e}  db {rel(b.grad.cpu(), bc.grad):.2e}  |db| {float(bc.grad.norm()):.3f} |dw| {float(wc.grad.norm()):.3f}")
