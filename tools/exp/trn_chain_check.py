#!/usr/bin/env python3
"""Temporal pooling chain with active dropout, bf16: every intermediate gradient of the HIP path against the storage model."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from egopack_amd import ops  # noqa: E402
from egopack_amd.models.temporal_pooling.trn_pooling import TRNPooling  # noqa: E402
from oracle import storage as S  # noqa: E402


def rel(a, b):
    return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp(min=1e-30))


torch.manual_seed(0)
for p_drop in (0.0, 0.5):
    trn = TRNPooling(1536, 1024, 3, hidden_size=1024, dropout=p_drop).cuda().train()
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in trn.state_dict().items()}
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(2048, 3, 1536, device="cuda", generator=g).bfloat16()
    R = torch.randn(2048, 1024, device="cuda", generator=g).bfloat16()
    p = trn.proj
    with ops.compute_mode("bf16"), ops.tap_dropout_masks() as masks:
        h0 = p[0](trn._rows(x)); y0 = p[1](h0, relu=True, p=p_drop); h1 = p[4](y0); y1 = p[5](h1, relu=True, p=p_drop); out = p[8](y1)
        hip = [h0, y0, h1, y1, out]
        for t in hip:
            t.retain_grad()
        out.backward(R)
        torch.cuda.synchronize()
    ms = [m.cpu().float() for m in masks] or [None, None]
    with S.bf16_storage():
        xc = x.float().cpu().reshape(2048, -1)
        a0 = S.act(F.linear(xc, S.weight(sd["proj.0.weight"]), sd["proj.0.bias"]))
        t0 = F.relu(F.layer_norm(a0, (1024,), sd["proj.1.weight"], sd["proj.1.bias"], 1e-5))
        b0 = S.act(t0 * ms[0] / (1 - p_drop) if ms[0] is not None else t0)
        a1 = S.act(F.linear(b0, S.weight(sd["proj.4.weight"]), sd["proj.4.bias"]))
        t1 = F.relu(F.layer_norm(a1, (1024,), sd["proj.5.weight"], sd["proj.5.bias"], 1e-5))
        b1 = S.act(t1 * ms[1] / (1 - p_drop) if ms[1] is not None else t1)
        o = S.act(F.linear(b1, S.weight(sd["proj.8.weight"]), sd["proj.8.bias"]))
        mod = [a0, b0, a1, b1, o]
        for t in mod:
            t.retain_grad()
        (o * R.float().cpu()).sum().backward()
    print(f"dropout {p_drop}")
    for name, a, b in zip(("h0", "y0", "h1", "y1", "out"), hip, mod):
        print(f"  {name}: value {rel(a.float().cpu(), b):.2e}  grad {rel(a.grad.float().cpu(), S._r(b.grad)):.2e}")
    for k, v in sd.items():
        print(f"  {k}: {rel(dict(trn.named_parameters())[k].grad.cpu(), v.grad):.2e}")
    # self-consistency of the second LayerNorm's backward on the HIP path's OWN tensors (host arithmetic in f64)
    h1c, dy = h1.detach().double().cpu(), y1.grad.double().cpu()
    w5, b5 = p[5].weight.detach().double().cpu(), p[5].bias.detach().double().cpu()
    mu, var = h1c.mean(1, keepdim=True), h1c.var(1, unbiased=False, keepdim=True)
    rs = 1 / (var + 1e-5).sqrt()
    xh = (h1c - mu) * rs
    gate = ((xh * w5 + b5) > 0).double()
    gg = dy * gate * (ms[1].double() / (1 - p_drop) if ms[1] is not None else 1.0)
    dxh = gg * w5
    dx_host = rs * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
    print(f"  self-consistency (host f64 on the HIP tensors): db5 {rel(p[5].bias.grad.cpu(), gg.sum(0)):.2e}  dw5 {rel(p[5].weight.grad.cpu(), (gg * xh).sum(0)):.2e}"
          f"  dx5 {rel(h1.grad.float().cpu(), dx_host.float().bfloat16().float()):.2e}")
    gm = (mod[3].grad.double()) # model grad at b1 (pre-round)
    print(f"  model: |db5| {float(sd['proj.5.bias'].grad.norm()):.4f} hip |db5| {float(p[5].bias.grad.norm()):.4f}; y1.grad vs model round(b1.grad) {rel(y1.grad.float().cpu(), S._r(mod[3].grad)):.2e}")
