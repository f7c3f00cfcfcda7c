"""Storage model of the HIP path's 'bf16' compute mode for the oracle -- TEST INFRASTRUCTURE ONLY.

The product stores every activation and every activation gradient as bf16 in HBM, reads the weights of a contraction from
their bf16 copies, and computes everything else (accumulators, statistics, losses, parameter gradients) in f32.  With this
model switched on (``with bf16_storage():``) the oracle rounds at exactly those points and nowhere else:

  act(x)     an activation the product stores: value rounded to bf16 forward, its gradient rounded to bf16 backward (the
             gradient of a stored activation is itself a stored tensor; several consumers' contributions are summed in f32
             first, as the product's fused epilogues do);
  weight(w)  a contraction's weight operand: bf16-rounded value forward, identity backward (weight gradients are f32);
  grad(x)    a tensor kept in f32 whose gradient is handed to a contraction as a bf16 operand (the logits).

Everything is the identity when the model is off (the default): the pinned f32 oracle is unchanged.  What this buys: against
the f32 oracle the bf16-mode gradients differ by 8-22 % (rounding at ~40 storage points plus the ReLU / max gates that the
rounding flips); against THIS model the same gradients must agree to accumulation order, so a wrong term can no longer hide
inside the rounding noise (tests/test_gpu_configs.py)."""
from __future__ import annotations

import torch

_on = {"bf16": False}


def _r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _r(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _r(g)


def act(x: torch.Tensor) -> torch.Tensor:
    return _RoundBoth.apply(x) if _on["bf16"] else x


def weight(w: torch.Tensor) -> torch.Tensor:
    return w + (_r(w) - w).detach() if _on["bf16"] else w


def grad(x: torch.Tensor) -> torch.Tensor:
    return _RoundGrad.apply(x) if _on["bf16"] else x


def is_on() -> bool:
    return _on["bf16"]


class bf16_storage:
    """``with bf16_storage():`` -- the oracle rounds where the product's 'bf16' mode stores bf16; ``bf16_storage(False)``
    switches the model off inside an enclosing scope (the f32-grade features behind the prototype search)."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _on["bf16"] = _on["bf16"], self.on

    def __exit__(self, *a):
        _on["bf16"] = self.prev
