"""GPU parity of every C-ABI kernel against the CPU oracle / plain torch fp32-fp64 references.

Tolerances (stated per test):
  * exact-f32 MFMA contractions and all non-GEMM kernels: rtol 1e-4 / atol 1e-5 (summation order);
  * bf16 MFMA contractions: compared with a reference computed from bf16-ROUNDED operands in fp64
    (the only error left is fp32 accumulation order): rtol 1e-3 / atol 1e-3;
  * integer / index outputs (arg-max, k-NN indices, masks, counts): bit-exact.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import pyg_ops as P  # noqa: E402
from oracle import path as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd import ops as _ops
    return _ops


DEV = "cuda"


def gen(seed):
    return torch.Generator().manual_seed(seed)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


# ---------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------
SHAPES = [(130, 70, 40), (257, 129, 144), (64, 7, 32), (33, 1, 32), (300, 256, 200), (128, 128, 64)]


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("transA,transB", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_layouts(ops, compute, transA, transB, M, N, K):
    g = gen(M * 1000 + N * 10 + K)
    A = torch.randn((K, M) if transA else (M, K), generator=g)
    B = torch.randn((K, N) if transB else (N, K), generator=g)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    opA = (A.t() if transA else A).double()
    opB = (B.t() if transB else B).double()
    if compute == "bf16":
        opA, opB = bf16_round(opA.float()), bf16_round(opB.float())
        tol = dict(rtol=1e-3, atol=1e-3)
    else:
        tol = dict(rtol=1e-4, atol=1e-4)
    ref = (opA @ opB.t() + bias.double()).float() + res
    out = torch.empty(M, N, device=DEV)
    Ad, Bd = A.to(DEV), B.to(DEV)
    ops.gemm(M, N, Ad, A.shape[1], Bd, B.shape[1], K, out, N, transA=transA, transB=transB, bias=bias.to(DEV),
             residual=res.to(DEV), ldr=N, compute=ops.BF16 if compute == "bf16" else ops.F32)
    torch.testing.assert_close(out.cpu(), ref, **tol)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_gemm_two_source_relu_accumulate(ops, compute):
    g = gen(7)
    M, N, K1, K2 = 150, 96, 40, 72
    A1, A2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    B1, B2 = torch.randn(N, K1, generator=g), torch.randn(N, K2, generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    rnd = bf16_round if compute == "bf16" else (lambda t: t.double())
    ref = torch.relu(0.5 * (rnd(A1) @ rnd(B1).t() + rnd(A2) @ rnd(B2).t()) + C0.double() + bias.double()).float()
    out = C0.clone().to(DEV)
    ops.gemm(M, N, A1.to(DEV), K1, B1.to(DEV), K1, K1, out, N, A2=A2.to(DEV), lda2=K2, B2=B2.to(DEV), ldb2=K2, K2=K2,
             bias=bias.to(DEV), act=1, accumulate=True, alpha=0.5, compute=ops.BF16 if compute == "bf16" else ops.F32)
    tol = dict(rtol=1e-3, atol=1e-3) if compute == "bf16" else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.cpu(), ref, **tol)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_gemm_splitk_matches_single_pass(ops, compute):
    """dW-shaped contraction (few tiles, deep K): the library picks split-K slabs; compare with fp64."""
    from egopack_amd import _lib
    g = gen(11)
    M, N, K = 128, 256, 8192
    c = ops.BF16 if compute == "bf16" else ops.F32
    assert _lib.load().egk_gemm_splitk(M, N, K, c) > 1
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)  # both transposed (dW form)
    C0 = torch.randn(M, N, generator=g)
    rnd = bf16_round if compute == "bf16" else (lambda t: t.double())
    ref = (rnd(A).t() @ rnd(B) + C0.double()).float()
    out = C0.clone().to(DEV)
    ops.gemm(M, N, A.to(DEV), M, B.to(DEV), N, K, out, N, transA=True, transB=True, accumulate=True, compute=c)
    tol = dict(rtol=2e-3, atol=5e-3) if compute == "bf16" else dict(rtol=1e-4, atol=5e-4)
    torch.testing.assert_close(out.cpu(), ref, **tol)
    out2 = C0.clone().to(DEV)
    ops.gemm(M, N, A.to(DEV), M, B.to(DEV), N, K, out2, N, transA=True, transB=True, accumulate=True, compute=c)
    assert torch.equal(out, out2)  # slab reduction is order-fixed: bitwise reproducible


def test_gemm_linearity_full_size(ops):
    """Size-independent property at the bench shape (M=6144, K=4608, N=1024): A.(B1+B2)^T == A.B1^T + A.B2^T
    on the exact-f32 path (no CPU reference needed)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 6144, 1024, 4608
    A = torch.randn(M, K, device=DEV, generator=g)
    B1 = torch.randn(N, K, device=DEV, generator=g)
    B2 = torch.randn(N, K, device=DEV, generator=g)
    o1, o2, o3 = (torch.empty(M, N, device=DEV) for _ in range(3))
    ops.gemm(M, N, A, K, B1, K, K, o1, N, compute=ops.F32)
    ops.gemm(M, N, A, K, B2, K, K, o2, N, compute=ops.F32)
    ops.gemm(M, N, A, K, B1 + B2, K, K, o3, N, compute=ops.F32)
    torch.testing.assert_close(o3, o1 + o2, rtol=1e-3, atol=2e-2)
    rows = torch.tensor([0, 17, 4095, 6143])
    ref = (A[rows].double().cpu() @ B1.double().cpu().t()).float()
    torch.testing.assert_close(o1[rows].cpu(), ref, rtol=1e-4, atol=2e-3)


# ---------------------------------------------------------------------------------------------------------
# autograd Linear
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("compute,tol", [("f32", 1e-4), ("bf16", 3e-2)])
def test_linear_autograd_two_source_residual(ops, compute, tol):
    g = gen(21)
    M, K1, K2, N = 70, 40, 32, 24
    x, x2 = torch.randn(M, K1, generator=g), torch.randn(M, K2, generator=g)
    W, W2, b = torch.randn(N, K1, generator=g), torch.randn(N, K2, generator=g), torch.randn(N, generator=g)
    r, w = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    cpu = [t.clone().requires_grad_(True) for t in (x, W, b, x2, W2, r)]
    ref = F.linear(cpu[0], cpu[1], cpu[2]) + F.linear(cpu[3], cpu[4]) + cpu[5]
    (ref * w).sum().backward()
    dev = [t.clone().to(DEV).requires_grad_(True) for t in (x, W, b, x2, W2, r)]
    with ops.compute_mode(compute):
        out = ops.linear(dev[0], dev[1], dev[2], x2=dev[3], W2=dev[4], residual=dev[5])
        (out * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=tol, atol=tol * 10)
    for a, c in zip(dev, cpu):
        torch.testing.assert_close(a.grad.cpu(), c.grad, rtol=tol, atol=tol * 20)


def test_linear_relu_and_fused_grad_slot(ops):
    g = gen(22)
    M, K, N = 50, 32, 40
    x, W, b, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    cx, cW, cb = (t.clone().requires_grad_(True) for t in (x, W, b))
    (torch.relu(F.linear(cx, cW, cb)) * w).sum().backward()
    dx, dW, db = (t.clone().to(DEV).requires_grad_(True) for t in (x, W, b))
    dW.grad = torch.full_like(dW, 2.0)  # pre-existing contiguous grad: backward accumulates into it in place
    db.grad = torch.zeros_like(db)
    slot = dW.grad
    with ops.compute_mode("f32"):
        out = ops.linear(dx, dW, db, relu=True)
        (out * w.to(DEV)).sum().backward()
    assert dW.grad is slot
    torch.testing.assert_close(dW.grad.cpu(), cW.grad + 2.0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.grad.cpu(), cb.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-4, atol=1e-4)


def test_multi_linear_matches_concatenation(ops):
    g = gen(23)
    xs = [torch.randn(m, 48, generator=g) for m in (30, 17, 64)]
    W, b = torch.randn(40, 48, generator=g), torch.randn(40, generator=g)
    cW, cb = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.linear(torch.cat(xs), cW, cb)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    dW, db = W.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    with ops.compute_mode("f32"):
        out = ops.multi_linear([x.to(DEV) for x in xs], dW, db)
        (out * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dW.grad.cpu(), cW.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.grad.cpu(), cb.grad, rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------
# normalisation
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,cols", [(37, 40), (64, 32), (130, 1024), (9, 4096), (5, 250), (130, 2048), (70, 3072), (1031, 4096), (3, 1280)])
@pytest.mark.parametrize("relu", [False, True])
def test_rowln_fwd_bwd(ops, rows, cols, relu):
    g = gen(rows * cols)
    x = torch.randn(rows, cols, generator=g) * 2 + 0.3
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    wt = torch.randn(rows, cols, generator=g)
    cx, cw, cb = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.layer_norm(cx, (cols,), cw, cb, 1e-5)
    ref = torch.relu(ref) if relu else ref
    (ref * wt).sum().backward()
    dx, dw, db = (t.clone().to(DEV).requires_grad_(True) for t in (x, w, b))
    out = ops.row_layernorm(dx, dw, db, 1e-5, relu=relu)
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dw.grad.cpu(), cw.grad, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(db.grad.cpu(), cb.grad, rtol=1e-3, atol=1e-3)


def test_rowln_dropout_mask_semantics(ops):
    """Dropout inside the fused LayerNorm launch: the oracle applied with the kernel's own keep-mask must
    reproduce output and gradients; the mask keeps ~ (1-p) of the elements and differs between calls."""
    g = gen(5)
    rows, cols, p = 64, 1024, 0.5
    x, w, b = torch.randn(rows, cols, generator=g), torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    wt = torch.randn(rows, cols, generator=g)
    ops.manual_seed(1234)
    dx = x.clone().to(DEV).requires_grad_(True)
    out = ops.row_layernorm(dx, w.to(DEV), b.to(DEV), 1e-5, relu=True, p=p, training=True)
    mask = ops.last_rowln_mask(out).cpu()
    assert mask.dtype == torch.uint8 and set(mask.unique().tolist()) <= {0, 1}
    assert abs(mask.float().mean().item() - (1 - p)) < 0.02
    cx = x.clone().requires_grad_(True)
    ref = torch.relu(F.layer_norm(cx, (cols,), w, b, 1e-5)) * mask.float() / (1 - p)
    (ref * wt).sum().backward()
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-3, atol=1e-4)
    out2 = ops.row_layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, relu=True, p=p, training=True)
    assert not torch.equal((out2 != 0), (out.detach() != 0))
    out3 = ops.row_layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, relu=True, p=p, training=False)
    torch.testing.assert_close(out3.cpu(), torch.relu(F.layer_norm(x, (cols,), w, b, 1e-5)), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows,cols,segs", [(40, 32, [0, 40]), (64, 1024, [0, 10, 64]), (300, 256, [0, 100, 101, 300])])
def test_graphln_lrelu_fwd_bwd(ops, rows, cols, segs):
    g = gen(rows + cols)
    x = torch.randn(rows, cols, generator=g) * 1.5 + 0.2
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    wt = torch.randn(rows, cols, generator=g)
    cx, cw, cb = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = torch.cat([F.leaky_relu(P.graph_layer_norm(cx[s:e], cw, cb), 0.2) for s, e in zip(segs[:-1], segs[1:])])
    (ref * wt).sum().backward()
    dx, dw, db = (t.clone().to(DEV).requires_grad_(True) for t in (x, w, b))
    seg = torch.tensor(segs, dtype=torch.int32, device=DEV)
    out = ops.graph_layernorm_lrelu(dx, dw, db, seg, 1e-5, 0.2)
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(dw.grad.cpu(), cw.grad, rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(db.grad.cpu(), cb.grad, rtol=1e-3, atol=2e-3)


def test_graphln_statistics_summed_with_a_second_rank(ops):
    """Exact cross-rank statistics (ops.set_graph_ln_exchange): this process plays rank A of two; the exchange function
    adds what rank B would contribute (computed here in f64 from B's rows).  Rank A's outputs and input gradients must be
    those of the oracle's graph LayerNorm over the UNION of the rows -- the single-process result on the global batch."""
    g = gen(77)
    cols, na, nb = 256, [40, 60], [24, 36]
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    xa = [torch.randn(n, cols, generator=g) * 1.5 + 0.3 for n in na]
    xb = [torch.randn(n, cols, generator=g) * 0.7 - 0.2 for n in nb]
    wa = [torch.randn(n, cols, generator=g) for n in na]
    wb = [torch.randn(n, cols, generator=g) for n in nb]
    # oracle on the union, segment by segment
    leaves = [torch.cat([a, bb]).clone().requires_grad_(True) for a, bb in zip(xa, xb)]
    cw, cb = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    outs = [F.leaky_relu(P.graph_layer_norm(l, cw, cb), 0.2) for l in leaves]
    sum((o * torch.cat([u, v])).sum() for o, u, v in zip(outs, wa, wb)).backward()
    # rank B's contributions, f64
    fwd_b = torch.tensor([[x.double().sum(), (x.double() ** 2).sum(), x.numel()] for x in xb], dtype=torch.float64)
    bwd_b = []
    for s in range(2):
        u = torch.cat([xa[s], xb[s]]).double()
        mean, std = u.mean(), u.std(unbiased=False)
        xhat = ((xb[s].double() - mean) / (std + 1e-5)).requires_grad_(True)
        (F.leaky_relu(xhat * w.double() + b.double(), 0.2) * wb[s].double()).sum().backward()
        bwd_b.append([xhat.grad.sum(), (xhat.grad * xhat.detach()).sum(), xb[s].numel()])
    bwd_b = torch.tensor(bwd_b, dtype=torch.float64)
    calls = []

    def exchange(buf):
        assert buf.dtype == torch.float64 and tuple(buf.shape) == (2, 3)
        buf += (fwd_b if not calls else bwd_b).to(buf.device)
        calls.append(1)
    prev = ops.set_graph_ln_exchange(exchange)
    try:
        x = torch.cat(xa).to(DEV).requires_grad_(True)
        dw, db = w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        seg = torch.tensor([0, na[0], na[0] + na[1]], dtype=torch.int32, device=DEV)
        out = ops.graph_layernorm_lrelu(x, dw, db, seg, 1e-5, 0.2)
        (out * torch.cat(wa).to(DEV)).sum().backward()
    finally:
        ops.set_graph_ln_exchange(prev)
    assert len(calls) == 2
    ref_out = torch.cat([o[:n] for o, n in zip(outs, na)]).detach()
    ref_dx = torch.cat([l.grad[:n] for l, n in zip(leaves, na)])
    torch.testing.assert_close(out.detach().cpu(), ref_out, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(x.grad.cpu(), ref_dx, rtol=1e-3, atol=1e-4)
    # and without the exchange the same call is a different (per-rank) normalisation
    plain = ops.graph_layernorm_lrelu(torch.cat(xa).to(DEV), w.to(DEV), b.to(DEV), seg, 1e-5, 0.2)
    assert (plain.cpu() - ref_out).abs().max() > 1e-2


def test_graphln_full_size_properties(ops):
    """[6144, 1024], 3 segments: each output segment (before the affine/LeakyReLU, w=1,b=0,slope=1) has
    mean 0 and population std 1/(1+eps/std) -- the defining property, checked on the GPU in fp64."""
    x = torch.randn(6144, 1024, device=DEV) * 3 + 1
    seg = torch.tensor([0, 2048, 4096, 6144], dtype=torch.int32, device=DEV)
    y = ops.graph_layernorm_lrelu(x, torch.ones(1024, device=DEV), torch.zeros(1024, device=DEV), seg, 1e-5, 1.0)
    for s in range(3):
        blk = y[s * 2048:(s + 1) * 2048].double()
        assert abs(blk.mean().item()) < 1e-5
        assert abs(blk.std(unbiased=False).item() - 1.0) < 1e-4


# ---------------------------------------------------------------------------------------------------------
# graph ops
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cols,dt", [(1024, torch.bfloat16), (250, torch.float32), (64, torch.float32), (2048, torch.bfloat16)])
def test_banded_gather_is_the_csr_gather_bit_for_bit(ops, cols, dt):
    """egk_csr_gather_banded (rows coded {i - 1, i, i + 1}: no index fetch) against egk_csr_gather on the same CSR: band
    sequences, LTA sequences (forecast nodes = general rows), isolated rows, self loops; the same sums in the same order."""
    from egopack_amd import data as D
    g = gen(cols)
    parts, n = [], 0
    for s_, T in enumerate((9, 32, 5, 1, 12)):
        parts.append(D.radius_band_edges(torch.arange(T), 1) + n)
        n += T
    y = torch.stack([torch.randint(1, 5, (14,), generator=g), torch.randint(0, 5, (14,), generator=g)], 1)
    y[:3] = -1
    parts.append(D.lta_connectivity_edges(torch.arange(14), y, 1.5) + n)
    n += 14 + 3  # + isolated rows
    parts.append(torch.tensor([[2, 40, 20], [2, 40, 20]]))  # three self loops
    ei = torch.cat(parts, 1)
    keep = torch.ones(ei.shape[1], dtype=torch.bool)
    keep[-1] = False
    order = torch.argsort(ei[1, keep] * n + ei[0, keep])  # rows in ascending-source order (a row of 3 adjacent entries: code 7) ...
    ei = torch.cat([ei[:, keep][:, order], ei[:, ~keep]], 1)  # ... but one self loop behind its row's other entries: a general row
    graph = D.build_csr(ei, n).to(DEV)
    assert int(graph.band[20]) == 0xFF
    assert (graph.band == 0xFF).any() and (graph.band == 5).any() and (graph.band == 0).any() and (graph.band == 7).any()
    x = torch.randn(n, cols, generator=g).to(DEV).to(dt)
    a, b = torch.empty_like(x), torch.empty_like(x)
    ops._csr_gather(x, graph.rowptr, graph.col, None, None, a, graph.heavy, graph.heavy_mode)
    ops._csr_gather(x, graph.rowptr, graph.col, None, None, b, graph.heavy, graph.heavy_mode, band=graph.band)
    assert torch.equal(a, b)
    ref = P.scatter_mean(x.float().cpu()[ei[0]], ei[1], n)
    torch.testing.assert_close(b.float().cpu(), ref, **(OUT16 if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-6)))


def test_frozen_weight_copies_follow_the_parameter(ops):
    """ops.weight_operand keeps ONE bf16 copy of a frozen (requires_grad = False) weight while the parameter is not
    written; an in-place write or a new storage makes the next call convert again."""
    W = torch.nn.Parameter(torch.randn(64, 128, device=DEV), requires_grad=False)
    a = ops.weight_operand(W, torch.bfloat16)
    assert ops.weight_operand(W, torch.bfloat16) is a and torch.equal(a.float(), W.detach().to(torch.bfloat16).float())
    with torch.no_grad():
        W.mul_(2.0)
    b = ops.weight_operand(W, torch.bfloat16)
    assert b is not a and torch.equal(b.float(), W.detach().to(torch.bfloat16).float())
    W.data = torch.randn(64, 128, device=DEV)
    c = ops.weight_operand(W, torch.bfloat16)
    assert c is not b and torch.equal(c.float(), W.detach().to(torch.bfloat16).float())
    T_ = torch.nn.Parameter(torch.randn(64, 128, device=DEV))  # trainable without an optimizer shadow: converted per call
    assert ops.weight_operand(T_, torch.bfloat16) is not ops.weight_operand(T_, torch.bfloat16)


def test_grouped_projection_infer_equals_the_heads_one_by_one(ops):
    """The detached auxiliary projections of the EgoPack step as three grouped launches (ops.grouped_projection_infer)
    against ProjectionTask.forward_features per head: identical bits (same kernels on the same operands)."""
    from egopack_amd.models.tasks import LTATask, PNRTask, RecognitionTask
    ops.set_compute("bf16")
    try:
        torch.manual_seed(3)
        tasks = [RecognitionTask(256, 256, (7, 11)), LTATask(256, 256, (7, 11)), PNRTask(256, 256)]
        for t in tasks:
            t.to(DEV).eval()
            for p in t.parameters():
                p.requires_grad_(False)
        x = torch.randn(192, 256, device=DEV).to(torch.bfloat16)
        got = ops.grouped_projection_infer(x, [t.net for t in tasks], out_f32=True)
        assert got is not None and all(g.dtype == torch.float32 for g in got)
        for t, g in zip(tasks, got):
            ref = t.forward_features(x, out_f32=True)
            torch.testing.assert_close(g, ref, rtol=1e-5, atol=1e-5)
    finally:
        ops.set_compute("f32")


def test_precise_grouped_projection_takes_its_operand_halves_from_the_layernorm_launch(ops, monkeypatch):
    """Inside a precise scope the grouped row LayerNorm of ``ops.grouped_projection_infer`` also stores the bf16 halves of its
    result (egk_tee_split_next -> egk_rowln_group_fwd) instead of a split launch of its own: the same bits as with the split
    launch (EGK_DISABLE=x3_tee), and close to the f32 evaluation of the heads (reference models/tasks/task.py:17-26)."""
    from egopack_amd.models.tasks import LTATask, PNRTask, RecognitionTask
    torch.manual_seed(5)
    tasks = [RecognitionTask(256, 256, (7, 11)), LTATask(256, 256, (7, 11)), PNRTask(256, 256)]
    for t in tasks:
        t.to(DEV).eval()
        for p in t.parameters():
            p.requires_grad_(False)
    x = torch.randn(136, 256, device=DEV)
    out = {}
    for tee in (True, False):
        if not tee:
            monkeypatch.setenv("EGK_DISABLE", "x3_tee")
        with ops.precise_scope():
            got = ops.grouped_projection_infer(x, [t.net for t in tasks], out_f32=True)
        assert got is not None
        out[tee] = [g.clone() for g in got]
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b)
    for t, g in zip(tasks, out[True]):
        h = torch.nn.functional.linear(x, t.net[1].weight, t.net[1].bias)
        h = torch.relu(torch.nn.functional.layer_norm(h, (h.shape[1],), t.net[2].weight, t.net[2].bias, t.net[2].eps))
        ref = torch.nn.functional.linear(h, t.net[4].weight, t.net[4].bias)
        torch.testing.assert_close(g, ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("rows,cols,f16", [(6144, 1024, True), (6144, 1024, False), (37, 260, True), (5, 4, True)])
def test_search_prep_launch_equals_its_three_passes(ops, rows, cols, f16):
    """egk_row_inv_norm_cast (the grouped prototype search's row norms, bf16 rounding and half rounding in one pass over the
    feature rows; reference models/graphONE/graphONE.py:119-151) against egk_row_inv_norm / egk_cast / egk_cast_f16: the same
    bits, values beyond the half range included."""
    from egopack_amd import _lib
    lib = _lib.load()
    x = torch.randn(rows, cols, generator=gen(rows + cols)).to(DEV)
    x[0, 0], x[rows - 1, cols - 1] = 1.0e5, -7.0e-6  # inf in half / a half subnormal
    inv = torch.empty(rows, dtype=torch.float32, device=DEV)
    hi = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    h16 = torch.empty_like(hi) if f16 else None
    rc = lib.egk_row_inv_norm_cast(ops._stream(), ops._p(x), ops._p(inv), ops._p(hi), ops._p(h16) if f16 else None, rows, cols)
    assert rc == 0, _lib.last_error()
    assert torch.equal(inv, ops.row_inv_norm(x))
    assert torch.equal(hi.view(torch.int16), ops.cast_raw(x, torch.bfloat16).view(torch.int16))
    if f16:
        assert torch.equal(h16.view(torch.int16), ops._cast_f16_bits(x).view(torch.int16))
        assert torch.equal(h16.view(torch.float16).float().cpu(), x.cpu().to(torch.float16).float())
    bad = lib.egk_row_inv_norm_cast(ops._stream(), ops._p(x), ops._p(inv), ops._p(hi), None, rows, cols - 1)
    assert bad != 0 and "multiple of 4" in _lib.last_error()


def test_pe_add(ops):
    g = gen(9)
    x = torch.randn(50, 64, generator=g)
    pos = torch.randint(-128, 128, (50,), generator=g)
    freq = P.positional_encoding_frequency(64)
    out = ops.pe_add(x.to(DEV), pos.to(DEV), freq.to(DEV))
    torch.testing.assert_close(out.cpu(), x + P.positional_encoding(pos, freq), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("cols", [32, 1024, 250])
def test_csr_mean_aggregate_fwd_bwd(ops, cols):
    from egopack_amd.data import build_csr, lta_connectivity_edges, radius_band_edges
    g = gen(cols)
    # two sequences: a band graph and an LTA graph (irregular degrees, isolated-free)
    e1 = radius_band_edges(torch.arange(9) - 4, 2)
    y = torch.stack([torch.randint(1, 5, (12,), generator=g), torch.randint(0, 5, (12,), generator=g)], 1)
    y[:2] = -1
    e2 = lta_connectivity_edges(torch.arange(12), y, 1.5) + 9
    extra_isolated = 3  # rows without in-edges -> 0
    n = 9 + 12 + extra_isolated
    ei = torch.cat([e1, e2], 1)
    x = torch.randn(n, cols, generator=g)
    wt = torch.randn(n, cols, generator=g)
    cx = x.clone().requires_grad_(True)
    ref = P.scatter_mean(cx.index_select(0, ei[0]), ei[1], n)
    (ref * wt).sum().backward()
    graph = build_csr(ei, n).to(DEV)
    dx = x.clone().to(DEV).requires_grad_(True)
    out = ops.csr_mean_aggregate(dx, graph)
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-5, atol=1e-6)
    assert torch.equal(out[-extra_isolated:].cpu(), torch.zeros(extra_isolated, cols))


def test_gather_max_fwd_bwd(ops):
    g = gen(31)
    N, K, H, k = 40, 37, 1024, 4
    f, bank = torch.randn(N, H, generator=g), torch.randn(K, H, generator=g)
    nn = torch.stack([torch.randperm(K, generator=g)[:k] for _ in range(N)])
    wt = torch.randn(N, H, generator=g)
    cf = f.clone().requires_grad_(True)
    stack = torch.cat([bank[nn], cf.unsqueeze(1)], 1)  # [N, k+1, H]
    ref = stack.max(1).values
    (ref * wt).sum().backward()
    df = f.clone().to(DEV).requires_grad_(True)
    out = ops.gather_max(df, bank.to(DEV), nn.to(DEV))
    (out * wt.to(DEV)).sum().backward()
    assert torch.equal(out.detach().cpu(), ref.detach())  # max is exact
    assert torch.equal(df.grad.cpu(), cf.grad)


@pytest.mark.parametrize("k,H", [(4, 1024), (4, 512), (4, 256), (8, 1024), (8, 256), (3, 1024), (4, 320)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_gather_max_of_several_tasks_in_one_launch_with_every_load_up_front(ops, k, H, dt):
    """egk_gather_max_group_fwd (the rows of G tasks, each with its own bank and neighbour lists, in one launch; k = 4 / 8 and widths
    in multiples of 256 on the kernel that requests every load of a row up front) against one generic launch per task
    (egk_gather_max_tune(0)): values and winners bit for bit, ties included (features copied from bank rows); k = 3 and width 320
    take the generic kernel inside the grouped entry point."""
    import ctypes as C
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(33)
    G, N, K = 3, 130, 57
    banks = [torch.randn(K, H, generator=g).to(DEV) for _ in range(G)]
    nns = [torch.stack([torch.randperm(K, generator=g)[:k] for _ in range(N)]).to(DEV) for _ in range(G)]
    f = torch.randn(G * N, H, generator=g)
    f[5] = banks[0][nns[0][5, 1]].cpu()  # exact ties with a prototype row: the prototype (earlier message) wins
    f = f.to(dt).to(DEV)
    dti = ops.BF16 if dt == torch.bfloat16 else ops.F32
    st = torch.cuda.current_stream().cuda_stream
    ptrs = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    def grouped():
        m, arg = torch.empty_like(f), torch.empty((G * N, H), dtype=torch.uint8, device=DEV)
        rc = lib.egk_gather_max_group_fwd(st, f.data_ptr(), ptrs(banks), ptrs(nns), G, m.data_ptr(), arg.data_ptr(), N, H, k, dti)
        assert rc == 0, _lib.last_error()
        return m, arg
    m1, a1 = grouped()
    prev = lib.egk_gather_max_tune(0)
    try:
        m0, a0 = torch.empty_like(f), torch.empty((G * N, H), dtype=torch.uint8, device=DEV)
        for i in range(G):
            rc = lib.egk_gather_max_fwd(st, f[i * N:].data_ptr(), banks[i].data_ptr(), nns[i].data_ptr(), m0[i * N:].data_ptr(),
                                        a0[i * N:].data_ptr(), N, H, k, dti)
            assert rc == 0, _lib.last_error()
        m2, a2 = grouped()  # (the grouped entry point on the generic kernel)
    finally:
        lib.egk_gather_max_tune(prev)
    torch.cuda.synchronize()
    assert torch.equal(m1, m0) and torch.equal(a1, a0) and torch.equal(m2, m0) and torch.equal(a2, a0)
    ref = torch.cat([torch.cat([banks[i][nns[i]], f[i * N:(i + 1) * N].float().unsqueeze(1)], 1).max(1).values for i in range(G)])
    assert torch.equal(m1.float(), ref.to(dt).float())
    assert int(a1[5].ne(k).sum()) > 0


def test_segment_max_fwd_bwd(ops):
    g = gen(32)
    ptr = torch.tensor([0, 4, 4, 9, 20], dtype=torch.int32)  # one empty sequence -> zeros
    x = torch.randn(20, 96, generator=g)
    wt = torch.randn(4, 96, generator=g)
    batch = torch.repeat_interleave(torch.arange(4), (ptr[1:] - ptr[:-1]).long())
    cx = x.clone().requires_grad_(True)
    ref = P.global_max_pool(cx, batch, 4)
    (ref * wt).sum().backward()
    dx = x.clone().to(DEV).requires_grad_(True)
    out = ops.segment_max(dx, ptr.to(DEV))
    (out * wt.to(DEV)).sum().backward()
    assert torch.equal(out.detach().cpu(), ref.detach())
    assert torch.equal(dx.grad.cpu(), cx.grad)


@pytest.mark.parametrize("N,K,H,k", [(14, 37, 32, 4), (64, 4096, 1024, 4), (33, 500, 256, 8)])
def test_cosine_topk_indices(ops, N, K, H, k):
    g = gen(N + K)
    f, bank = torch.randn(N, H, generator=g), torch.randn(K, H, generator=g)
    edges, closest = O.compute_edges(f, bank, k)
    nn = ops.cosine_topk(f.to(DEV), bank.to(DEV), k).cpu()
    dist = O.cos_dissimilarity(f, bank)
    srt = dist.sort(dim=-1).values
    gap = (srt[:, 1:k + 1] - srt[:, :k]).min(dim=1).values  # smallest gap among the first k+1 distances
    safe = gap > 1e-5  # rows whose ranking cannot flip under fp32 summation-order noise
    assert safe.float().mean() > 0.9
    assert torch.equal(nn[safe], closest[safe])  # index op: exact wherever the ranking is well defined
    picked = torch.gather(dist, 1, nn)  # everywhere: the selected distances are the k smallest up to noise
    torch.testing.assert_close(picked, srt[:, :k], rtol=0, atol=2e-5)


def test_topk_tie_breaks_to_lower_index(ops):
    bank = torch.tensor([[1., 0.], [1., 0.], [0., 1.], [1., 0.]])
    f = torch.tensor([[2., 0.]])
    nn = ops.cosine_topk(f.to(DEV), bank.to(DEV), 3).cpu()
    assert nn.tolist() == [[0, 1, 3]]


@pytest.mark.parametrize("where", ["host", "device"])
@pytest.mark.parametrize("cols", [1024, 37])
def test_scatter_add_rows_f64(ops, where, cols):
    """Prototype accumulation (reference graphone.py:53): rows of one label summed in fp32 in node order (what the
    reference's per-batch fp32 scatter does on the CPU), the sum added to the fp64 bank row -- BIT-EXACT against that
    restatement, twice in a row (a second batch accumulates), with labels grouped on the host or on the device."""
    g = gen(41)
    rows, L = 300, 35
    bank = torch.zeros(L, cols, dtype=torch.float64, device=DEV)
    count = torch.zeros(L, dtype=torch.int64, device=DEV)
    ref = torch.zeros(L, cols, dtype=torch.float64)
    ref_count = torch.zeros(L, dtype=torch.int64)
    for it in range(2):
        x = torch.randn(rows, cols, generator=g)
        label = torch.randint(-1, L, (rows,), generator=g)
        label[:40] = 7  # a heavy label: 40+ rows in one group
        ops.scatter_add_rows_f64(x.to(DEV), label if where == "host" else label.to(DEV), bank, count)
        keep = label >= 0
        ref = ref + P.scatter_sum(x[keep], label[keep], L)  # fp64 + fp32, as graphone.py:53
        ref_count += torch.bincount(label[keep], minlength=L)
    assert torch.equal(bank.cpu(), ref)
    assert torch.equal(count.cpu(), ref_count)


def test_scatter_add_rows_f64_is_reproducible_and_handles_empty(ops):
    g = gen(43)
    x = torch.randn(500, 256, generator=g).to(DEV)
    label = torch.randint(0, 9, (500,), generator=g).to(DEV)
    outs = []
    for _ in range(2):
        bank = torch.zeros(9, 256, dtype=torch.float64, device=DEV)
        ops.scatter_add_rows_f64(x, label, bank, None)
        outs.append(bank.clone())
    assert torch.equal(outs[0], outs[1])
    bank = torch.zeros(9, 256, dtype=torch.float64, device=DEV)
    count = torch.zeros(9, dtype=torch.int64, device=DEV)
    ops.scatter_add_rows_f64(x, torch.full((500,), -1, dtype=torch.int64), bank, count)  # nothing labelled
    assert float(bank.abs().max()) == 0.0 and int(count.sum()) == 0


@pytest.mark.parametrize("K,H,k", [(37, 32, 4), (600, 1024, 8)])
def test_l2_topk_matches_cdist(ops, K, H, k):
    """distance_func='l2' (reference graphONE.py:126-127: cdist / 4096, argsort): indices exact wherever the ranking gap
    exceeds fp32 noise, and the selected distances are the k smallest everywhere."""
    g = gen(K + k)
    f, bank = torch.randn(50, H, generator=g), torch.randn(K, H, generator=g)
    nn = ops.nearest_prototypes(f.to(DEV), bank.to(DEV), k, "l2").cpu()
    _, closest = O.compute_edges(f, bank, k, "l2")
    dist = torch.cdist(f.double(), bank.double()) / 4096
    srt = dist.sort(dim=-1).values
    gap = (srt[:, 1:k + 1] - srt[:, :k]).min(dim=1).values
    safe = gap > 1e-8 * 4  # (distances are O(sqrt(2H)/4096) ~ 1e-2; fp32 noise on them ~1e-9)
    assert safe.float().mean() > 0.9
    assert torch.equal(nn[safe], closest[safe])
    torch.testing.assert_close(torch.gather(dist, 1, nn), srt[:, :k], rtol=0, atol=1e-8)
    with pytest.raises(ValueError):
        ops.nearest_prototypes(f.to(DEV), bank.to(DEV), k, "manhattan")


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gather_max_trainable_bank_gradient(ops, dt):
    """GraphONE(freeze=False): the prototype rows receive, through the max aggregation, the gradient of every element
    they won -- against torch autograd on the same values (fp64), exact selection."""
    g = gen(77)
    N, K, H, k = 70, 19, 260, 4
    f = torch.randn(N, H, generator=g).to(dt)
    bank = torch.randn(K, H, generator=g)
    nn = torch.stack([torch.randperm(K, generator=g)[:k] for _ in range(N)])
    dm = torch.randn(N, H, generator=g).to(dt)
    fd, bd = f.to(DEV).requires_grad_(True), bank.to(DEV).requires_grad_(True)
    m = ops.gather_max(fd, bd, nn.to(DEV))
    m.backward(dm.to(DEV))
    f64, b64 = f.double().requires_grad_(True), bank.double().requires_grad_(True)
    cand = torch.cat([b64[nn], f64[:, None]], 1)  # prototype messages first, the self loop last; first maximum wins
    ref, _ = cand.max(dim=1)
    ref.backward(dm.double())
    tol = dict(rtol=1e-6, atol=1e-6) if dt == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(m.detach().double().cpu(), ref.detach(), **(tol if dt == torch.float32 else dict(rtol=1e-2, atol=1e-2)))
    torch.testing.assert_close(fd.grad.double().cpu(), f64.grad, **tol)
    assert bd.grad.dtype == torch.float32
    torch.testing.assert_close(bd.grad.double().cpu(), b64.grad, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------
# losses / optimiser
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("smoothing", [0.0, 0.1])
def test_cross_entropy_heads_ignore_index(ops, smoothing):
    g = gen(51)
    N = 77
    l1, l2 = torch.randn(N, 115, generator=g) * 3, torch.randn(N, 478, generator=g) * 3
    y = torch.stack([torch.randint(0, 115, (N,), generator=g), torch.randint(0, 478, (N,), generator=g)], 1)
    y[::3] = -1
    wt = torch.randn(N, generator=g)
    c1, c2 = l1.clone().requires_grad_(True), l2.clone().requires_grad_(True)
    ref = (F.cross_entropy(c1, y[:, 0], ignore_index=-1, reduction="none", label_smoothing=smoothing)
           + F.cross_entropy(c2, y[:, 1], ignore_index=-1, reduction="none", label_smoothing=smoothing))
    (ref * wt).sum().backward()
    d1, d2 = l1.clone().to(DEV).requires_grad_(True), l2.clone().to(DEV).requires_grad_(True)
    out = ops.cross_entropy((d1, d2), y.to(DEV), smoothing)
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d1.grad.cpu(), c1.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(d2.grad.cpu(), c2.grad, rtol=1e-4, atol=1e-6)
    assert torch.equal(out.detach().cpu()[::3], torch.zeros(len(range(0, N, 3))))


def test_cross_entropy_single_head_1d_targets(ops):
    g = gen(52)
    l = torch.randn(9, 2, generator=g)
    y = torch.tensor([0, 1, -1, 1, 0, 0, 1, -1, 1])
    out = ops.cross_entropy(l.to(DEV), y.to(DEV), 0.1)
    ref = F.cross_entropy(l, y, ignore_index=-1, reduction="none", label_smoothing=0.1)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


def test_bce_with_logits(ops):
    g = gen(53)
    x = torch.randn(333, generator=g) * 4
    y = torch.randint(0, 2, (333,), generator=g)
    wt = torch.randn(333, generator=g)
    cx = x.clone().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(cx, y.float(), reduction="none")
    (ref * wt).sum().backward()
    dx = x.clone().to(DEV).requires_grad_(True)
    out = ops.bce_with_logits(dx, y.to(DEV))
    (out * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dx.grad.cpu(), cx.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,cols,mode", [(333, 256, "f32"), (2048, 1024, "f32"), (2048, 1024, "bf16"), (77, 1000, "bf16")])
def test_one_logit_head_with_bce_in_one_row_pass(ops, rows, cols, mode):
    """ops.linear1_bce = BCEWithLogits(reduction='none')(Linear(H, 1)(f).squeeze(), y.float()) -- models/tasks/pnr.py:37-52,
    main_temporal.py:117-121 -- plus d f, d W, d b for the announced backward seed, against torch on the same (rounded)
    operands, and against the contraction path of this library (the path it replaces)."""
    g = gen(rows + cols)
    f = torch.randn(rows, cols, generator=g)
    W, b = torch.randn(1, cols, generator=g) * 0.05, torch.randn(1, generator=g)
    y = torch.randint(0, 2, (rows,), generator=g)
    seed = 0.7 / rows
    ops.set_compute(mode)
    try:
        if mode == "bf16":
            f, Wr = r16(f), r16(W)
        else:
            Wr = W
        cf, cW, cb = f.clone().requires_grad_(True), Wr.clone().requires_grad_(True), b.clone().requires_grad_(True)
        z = (cf @ cW.t()).squeeze(1) + cb
        ref = F.binary_cross_entropy_with_logits(z, y.float(), reduction="none")
        ref.backward(torch.full_like(ref, seed))
        df = f.to(DEV).to(ops.act_dtype()).requires_grad_(True)
        dW, db = W.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        with ops.loss_seed(seed):
            assert ops.linear1_bce_ok(df, dW)
            loss, logits = ops.linear1_bce(df, dW, db, y.to(DEV))
        loss.backward(torch.full_like(loss, seed))
        tol = dict(rtol=2e-5, atol=2e-6) if mode == "f32" else dict(rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(logits.cpu(), z.detach(), rtol=1e-4 if mode == "f32" else 1e-2, atol=1e-4 if mode == "f32" else 2e-2)
        torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-4 if mode == "f32" else 1e-2, atol=1e-4 if mode == "f32" else 2e-2)
        gscale = float(cf.grad.abs().max())
        assert (df.grad.float().cpu() - cf.grad).abs().max() <= (1e-5 if mode == "f32" else 1.5e-2) * gscale
        wscale = float(cW.grad.abs().max())
        assert (dW.grad.cpu() - cW.grad).abs().max() <= (2e-5 if mode == "f32" else 1.5e-2) * wscale
        assert abs(float(db.grad.cpu()) - float(cb.grad)) <= (2e-5 if mode == "f32" else 1e-2) * max(1.0, abs(float(cb.grad)) * 100)
        # the path it replaces: classifier contraction + egk_bce_* -- same values up to summation order
        df2 = f.to(DEV).to(ops.act_dtype()).requires_grad_(True)
        dW2, db2 = W.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        z2 = ops.linear(df2, dW2, db2, out_f32=True).squeeze(1)
        l2 = ops.bce_with_logits(z2, y.to(DEV))
        l2.backward(torch.full_like(l2, seed))
        torch.testing.assert_close(loss.detach(), l2.detach(), **(dict(rtol=1e-4, atol=1e-5) if mode == "f32" else dict(rtol=5e-3, atol=5e-3)))
        assert (df.grad.float() - df2.grad.float()).abs().max().item() <= (1e-5 if mode == "f32" else 1e-2) * gscale
        assert (dW.grad - dW2.grad).abs().max().item() <= (2e-5 if mode == "f32" else 1e-2) * wscale
    finally:
        ops.set_compute("f32")


@pytest.mark.parametrize("rows,cols,mode,smoothing", [(16, 1024, "bf16", 0.0), (64, 1024, "bf16", 0.0), (16, 1024, "f32", 0.0),
                                                      (37, 264, "f32", 0.1), (256, 1024, "bf16", 0.1)])
def test_two_logit_head_with_cross_entropy_in_one_launch(ops, rows, cols, mode, smoothing):
    """ops.linear2_ce = CrossEntropy(reduction='none', ignore_index=-1, label_smoothing)(Linear(H, 2)(f), y) -- the OSCC head behind
    its max pool, models/tasks/oscc.py:65-79 + main_temporal.py:291 -- plus d f, d W, d b for the announced backward seed, in ONE
    launch: against torch on the same (rounded) operands, with ignored rows, and against the contraction path of this library
    (the eleven launches it replaces)."""
    g = gen(rows + cols + int(smoothing * 10))
    f = torch.randn(rows, cols, generator=g)
    W, b = torch.randn(2, cols, generator=g) * 0.05, torch.randn(2, generator=g)
    y = torch.randint(0, 2, (rows,), generator=g)
    y[::5] = -1  # ignored sequences: loss 0, no gradient, still counted by the caller's mean
    seed = 0.7 / rows
    ops.set_compute(mode)
    try:
        if mode == "bf16":
            f, Wr = r16(f), r16(W)
        else:
            Wr = W
        cf, cW, cb = f.clone().requires_grad_(True), Wr.clone().requires_grad_(True), b.clone().requires_grad_(True)
        z = cf @ cW.t() + cb
        ref = F.cross_entropy(z, y, reduction="none", ignore_index=-1, label_smoothing=smoothing)
        ref.backward(torch.full_like(ref, seed))
        df = f.to(DEV).to(ops.act_dtype()).requires_grad_(True)
        dW, db = W.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        with ops.loss_seed(seed):
            assert ops.linear2_ce_ok(rows, df, dW)
            loss, logits = ops.linear2_ce(df, dW, db, y.to(DEV), smoothing)
        loss.backward(torch.full_like(loss, seed))
        lt = dict(rtol=1e-4, atol=1e-4) if mode == "f32" else dict(rtol=1e-2, atol=2e-2)
        torch.testing.assert_close(logits.cpu(), z.detach(), **lt)
        torch.testing.assert_close(loss.detach().cpu(), ref.detach(), **lt)
        assert float(loss.detach()[::5].abs().max()) == 0.0 and float(df.grad[::5].float().abs().max()) == 0.0
        gscale, wscale = float(cf.grad.abs().max()), float(cW.grad.abs().max())
        assert (df.grad.float().cpu() - cf.grad).abs().max() <= (1e-5 if mode == "f32" else 1.5e-2) * gscale
        assert (dW.grad.cpu() - cW.grad).abs().max() <= (2e-5 if mode == "f32" else 1.5e-2) * wscale
        assert (db.grad.cpu() - cb.grad).abs().max() <= (2e-5 if mode == "f32" else 1e-2) * max(1.0, float(cb.grad.abs().max()) * 100)
        # the path it replaces: classifier contraction + cross entropy kernels -- same values up to summation order
        df2 = f.to(DEV).to(ops.act_dtype()).requires_grad_(True)
        dW2, db2 = W.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
        z2 = ops.linear(df2, dW2, db2, out_f32=True)
        l2 = ops.cross_entropy(z2, y.to(DEV), smoothing)
        l2.backward(torch.full_like(l2, seed))
        torch.testing.assert_close(loss.detach(), l2.detach(), **(dict(rtol=1e-4, atol=1e-5) if mode == "f32" else dict(rtol=5e-3, atol=5e-3)))
        assert (df.grad.float() - df2.grad.float()).abs().max().item() <= (1e-5 if mode == "f32" else 1e-2) * gscale
        assert (dW.grad - dW2.grad).abs().max().item() <= (2e-5 if mode == "f32" else 1e-2) * wscale
        with ops.loss_seed(seed):  # too many rows for one workgroup: the caller keeps the contraction path
            assert not ops.linear2_ce_ok(257, df, dW)
    finally:
        ops.set_compute("f32")


@pytest.mark.parametrize("n_src,average,mode", [(4, True, "bf16"), (4, False, "f32"), (2, True, "f32"), (3, True, "bf16")])
def test_two_logit_heads_of_several_sources_in_one_launch(ops, n_src, average, mode):
    """ops.linear2_ce_multi: the EgoPack OSCC head -- the primary pooled features and one pooled GraphONE feature per auxiliary task,
    each through its own Linear(H, 2), fused as stack(...).mean(0) / .sum(0), cross entropy with label smoothing 0.1
    (models/tasks/oscc.py:65-96) -- loss AND every gradient in one launch, against torch on the same (rounded) operands; a
    source that wants no gradient gets none."""
    rows, cols, smoothing = 64, 1024, 0.1
    g = gen(n_src * 7 + int(average))
    fs = [torch.randn(rows, cols, generator=g) for _ in range(n_src)]
    Ws = [torch.randn(2, cols, generator=g) * 0.05 for _ in range(n_src)]
    bs = [torch.randn(2, generator=g) for _ in range(n_src)]
    y = torch.randint(0, 2, (rows,), generator=g)
    y[3::7] = -1
    seed = 1.3 / rows
    ops.set_compute(mode)
    try:
        if mode == "bf16":
            fs, Wr = [r16(f) for f in fs], [r16(W) for W in Ws]
        else:
            Wr = Ws
        cf = [f.clone().requires_grad_(True) for f in fs]
        cW, cb = [W.clone().requires_grad_(True) for W in Wr], [b.clone().requires_grad_(True) for b in bs]
        zs = torch.stack([f @ W.t() + b for f, W, b in zip(cf, cW, cb)])
        z = zs.mean(0) if average else zs.sum(0)
        ref = F.cross_entropy(z, y, reduction="none", ignore_index=-1, label_smoothing=smoothing)
        ref.backward(torch.full_like(ref, seed))
        dfs = [f.to(DEV).to(ops.act_dtype()).requires_grad_(k != 1) for k, f in enumerate(fs)]  # (source 1: no gradient wanted)
        dWs, dbs = [W.clone().to(DEV).requires_grad_(True) for W in Ws], [b.clone().to(DEV).requires_grad_(True) for b in bs]
        with ops.loss_seed(seed):
            loss, logits = ops.linear2_ce_multi(dfs, dWs, dbs, y.to(DEV), smoothing, average=average)
        loss.backward(torch.full_like(loss, seed))
        lt = dict(rtol=1e-4, atol=1e-4) if mode == "f32" else dict(rtol=1e-2, atol=3e-2)
        torch.testing.assert_close(logits.cpu(), z.detach(), **lt)
        torch.testing.assert_close(loss.detach().cpu(), ref.detach(), **lt)
        assert dfs[1].grad is None
        for k in range(n_src):
            gs, ws_ = float(cf[k].grad.abs().max()), float(cW[k].grad.abs().max())
            if k != 1:
                assert (dfs[k].grad.float().cpu() - cf[k].grad).abs().max() <= (1e-5 if mode == "f32" else 1.5e-2) * gs
            assert (dWs[k].grad.cpu() - cW[k].grad).abs().max() <= (3e-5 if mode == "f32" else 1.5e-2) * ws_
            assert (dbs[k].grad.cpu() - cb[k].grad).abs().max() <= (3e-5 if mode == "f32" else 1e-2) * max(1.0, float(cb[k].grad.abs().max()) * 100)
    finally:
        ops.set_compute("f32")


def test_weighted_mean_sum_and_sum_tensors(ops):
    g = gen(54)
    a, b = torch.randn(100, generator=g), torch.randn(37, generator=g)
    ca, cb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = 0.5 * ca.mean() + 2.0 * cb.mean()
    ref.backward()
    da, db = a.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ops.weighted_mean_sum([da, db], [0.5, 2.0])
    out.backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(da.grad.cpu(), ca.grad)
    torch.testing.assert_close(db.grad.cpu(), cb.grad)
    ts = [torch.randn(13, 7, generator=g) for _ in range(4)]
    for scale in (1.0, 0.25):
        dts = [t.clone().to(DEV).requires_grad_(True) for t in ts]
        s = ops.sum_tensors(dts, scale)
        torch.testing.assert_close(s.detach().cpu(), torch.stack(ts).sum(0) * scale, rtol=1e-6, atol=1e-6)
        s.sum().backward()
        torch.testing.assert_close(dts[2].grad.cpu(), torch.full((13, 7), scale))


def test_dropout_op(ops):
    x = torch.ones(1 << 16, device=DEV, requires_grad=True)
    ops.manual_seed(7)
    y = ops.dropout(x, 0.25, True)
    kept = (y != 0).float().mean().item()
    assert abs(kept - 0.75) < 0.01
    torch.testing.assert_close(y[y != 0], torch.full_like(y[y != 0], 1 / 0.75))
    y.sum().backward()
    assert torch.equal((x.grad != 0), (y != 0))
    assert ops.dropout(x, 0.25, False) is x


def test_flat_adam_matches_torch_adam(ops):
    from egopack_amd.optim import FlatAdam
    g = gen(61)
    shapes = [(33, 7), (5,), (64, 64), (3,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(3)]
    cpu = [p.clone().requires_grad_(True) for p in ps]
    unused_cpu = torch.randn(4, generator=g).requires_grad_(True)
    ref = torch.optim.Adam(cpu + [unused_cpu], lr=1e-2, weight_decay=1e-3)
    dev = [p.clone().to(DEV).requires_grad_(True) for p in ps]
    unused = unused_cpu.detach().clone().to(DEV).requires_grad_(True)
    opt = FlatAdam(dev + [unused], lr=1e-2, weight_decay=1e-3)
    for it in range(3):
        ref.zero_grad()
        opt.zero_grad()
        for c, d, gr in zip(cpu, dev, grads[it]):
            c.grad = gr.clone()
            if d.grad is None:
                d.grad = gr.clone().to(DEV)
            else:
                d.grad.copy_(gr)
        ref.step()
        opt.step()
        if it == 0:
            assert all(d.data.data_ptr() >= opt.flat_p.data_ptr() for d in dev)  # re-homed into the flat buffer
    for c, d in zip(cpu, dev):
        torch.testing.assert_close(d.detach().cpu(), c.detach(), rtol=1e-5, atol=1e-6)
    assert torch.equal(unused.detach().cpu(), unused_cpu.detach())  # grad None -> skipped, as torch does


def test_ops_reject_cpu_tensors(ops):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.randn(4, 4), torch.randn(4, 4))


# ---------------------------------------------------------------------------------------------------------
# bf16 activations / operands in memory (mode 'bf16'): every kernel computes in f32 from bf16-rounded inputs,
# so the reference is the f32/f64 op applied to the ROUNDED inputs; the only extra error is the final rounding
# of a bf16 output (relative 2^-8) -> rtol 8e-3.
# ---------------------------------------------------------------------------------------------------------
BF = torch.bfloat16
OUT16 = dict(rtol=8e-3, atol=8e-3)


def r16(t):
    return t.to(BF).float()


@pytest.mark.parametrize("transA,transB", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(130, 70, 40), (257, 129, 144), (64, 7, 32), (300, 256, 200), (128, 128, 64)])
@pytest.mark.parametrize("out16", [False, True])
def test_gemm_bf16_memory_operands(ops, transA, transB, M, N, K, out16):
    g = gen(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if transA else (M, K), generator=g)
    B = torch.randn((K, N) if transB else (N, K), generator=g)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    opA = (A.t() if transA else A).to(BF).double()
    opB = (B.t() if transB else B).to(BF).double()
    ref = (opA @ opB.t() + bias.double()).float() + r16(res)
    out = torch.empty(M, N, device=DEV, dtype=BF if out16 else torch.float32)
    ops.gemm(M, N, A.to(DEV).to(BF), A.shape[1], B.to(DEV).to(BF), B.shape[1], K, out, N, transA=transA, transB=transB,
             bias=bias.to(DEV), residual=res.to(DEV).to(BF), ldr=N)
    torch.testing.assert_close(out.float().cpu(), ref, **(OUT16 if out16 else dict(rtol=1e-3, atol=1e-3)))


def test_gemm_bf16_two_source_splitk_accumulate(ops):
    g = gen(77)
    M, N, K1, K2 = 128, 256, 4096, 4096  # dW shape: few tiles, deep K -> split-K slabs
    A1, A2 = torch.randn(K1, M, generator=g), torch.randn(K2, M, generator=g)
    B1, B2 = torch.randn(K1, N, generator=g), torch.randn(K2, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = (A1.to(BF).double().t() @ B1.to(BF).double() + A2.to(BF).double().t() @ B2.to(BF).double() + C0.double()).float()
    out = C0.clone().to(DEV)
    ops.gemm(M, N, A1.to(DEV).to(BF), M, B1.to(DEV).to(BF), N, K1, out, N, A2=A2.to(DEV).to(BF), lda2=M,
             B2=B2.to(DEV).to(BF), ldb2=N, K2=K2, transA=True, transB=True, accumulate=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-3, atol=1e-2)


def test_cast_roundtrip(ops):
    x = torch.randn(1000, 37, device=DEV)
    h = ops.cast_raw(x, BF)
    assert h.dtype == BF and torch.equal(h, x.to(BF))
    assert torch.equal(ops.cast_raw(h, torch.float32), h.float())


def test_linear_autograd_full_bf16(ops):
    g = gen(88)
    M, K1, K2, N = 70, 40, 32, 24
    x, x2 = r16(torch.randn(M, K1, generator=g)), r16(torch.randn(M, K2, generator=g))
    W, W2, b = r16(torch.randn(N, K1, generator=g)), r16(torch.randn(N, K2, generator=g)), torch.randn(N, generator=g)
    r, w = r16(torch.randn(M, N, generator=g)), r16(torch.randn(M, N, generator=g))
    cpu = [t.clone().requires_grad_(True) for t in (x, W, b, x2, W2, r)]
    ref = torch.nn.functional.linear(cpu[0], cpu[1], cpu[2]) + torch.nn.functional.linear(cpu[3], cpu[4]) + cpu[5]
    (ref * w).sum().backward()
    dx, dx2, dr = (t.clone().to(DEV).to(BF).requires_grad_(True) for t in (x, x2, r))
    dW, db, dW2 = (t.clone().to(DEV).requires_grad_(True) for t in (W, b, W2))
    with ops.compute_mode("bf16"):
        out = ops.linear(dx, dW, db, x2=dx2, W2=dW2, residual=dr)
        assert out.dtype == BF
        (out.float() * w.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().float().cpu(), ref.detach(), rtol=1e-2, atol=3e-2)
    for a, c in zip((dx, dW, db, dx2, dW2, dr), cpu):
        torch.testing.assert_close(a.grad.float().cpu(), c.grad, rtol=2e-2, atol=6e-2)


@pytest.mark.parametrize("rows,cols", [(37, 40), (130, 1024), (9, 4096), (515, 4096), (66, 2048)])
def test_rowln_bf16_activations(ops, rows, cols):
    g = gen(rows + cols + 1)
    x = r16(torch.randn(rows, cols, generator=g) * 2 + 0.3)
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    wt = r16(torch.randn(rows, cols, generator=g))
    cx, cw, cb = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = torch.relu(F.layer_norm(cx, (cols,), cw, cb, 1e-5))
    (ref * wt).sum().backward()
    dx = x.clone().to(DEV).to(BF).requires_grad_(True)
    dw, db = w.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ops.row_layernorm(dx, dw, db, 1e-5, relu=True)
    assert out.dtype == BF
    (out.float() * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(out.detach().float().cpu(), ref.detach(), **OUT16)
    # the incoming gradient passes through the bf16 output of ``out.float()``'s backward: rounded once more
    torch.testing.assert_close(dx.grad.float().cpu(), cx.grad, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(dw.grad.cpu(), cw.grad, rtol=2e-2, atol=3e-2 * rows ** 0.5)
    torch.testing.assert_close(db.grad.cpu(), cb.grad, rtol=2e-2, atol=3e-2 * rows ** 0.5)


def test_graphln_csr_pe_bf16_activations(ops):
    from egopack_amd.data import build_csr, radius_band_edges
    g = gen(99)
    rows, cols = 64, 1024
    x = r16(torch.randn(rows, cols, generator=g) * 1.5 + 0.2)
    w, b = torch.randn(cols, generator=g), torch.randn(cols, generator=g)
    seg = [0, 24, 64]
    ref = torch.cat([F.leaky_relu(P.graph_layer_norm(x[s:e], w, b), 0.2) for s, e in zip(seg[:-1], seg[1:])])
    xd = x.to(DEV).to(BF)
    out = ops.graph_layernorm_lrelu(xd, w.to(DEV), b.to(DEV), torch.tensor(seg, dtype=torch.int32, device=DEV))
    assert out.dtype == BF
    torch.testing.assert_close(out.float().cpu(), ref, **OUT16)
    ei = torch.cat([radius_band_edges(torch.arange(32), 1), radius_band_edges(torch.arange(32), 2) + 32], 1)
    agg = ops.csr_mean_aggregate(xd, build_csr(ei, rows).to(DEV))
    torch.testing.assert_close(agg.float().cpu(), P.scatter_mean(x[ei[0]], ei[1], rows), **OUT16)
    pos = torch.randint(-64, 64, (rows,), generator=g)
    freq = P.positional_encoding_frequency(cols)
    pe = ops.pe_add(xd, pos.to(DEV), freq.to(DEV))
    torch.testing.assert_close(pe.float().cpu(), x + P.positional_encoding(pos, freq), **OUT16)


def test_gather_segmax_dropout_relu_bf16_activations(ops):
    g = gen(111)
    N, K, H, k = 40, 37, 256, 4
    f, bank = r16(torch.randn(N, H, generator=g)), torch.randn(K, H, generator=g)
    nn = torch.stack([torch.randperm(K, generator=g)[:k] for _ in range(N)])
    fd = f.to(DEV).to(BF).requires_grad_(True)
    m = ops.gather_max(fd, bank.to(DEV), nn.to(DEV))
    ref = torch.cat([bank[nn], f.unsqueeze(1)], 1).max(1).values
    torch.testing.assert_close(m.detach().float().cpu(), ref, **OUT16)  # bank values are rounded on output
    m.float().sum().backward()
    assert set(fd.grad.float().unique().tolist()) <= {0.0, 1.0}
    ptr = torch.tensor([0, 10, 40], dtype=torch.int32)
    sm = ops.segment_max(f.to(DEV).to(BF), ptr.to(DEV))
    assert torch.equal(sm.float().cpu(), torch.stack([f[:10].max(0).values, f[10:].max(0).values]))
    ops.manual_seed(3)
    y = ops.dropout(torch.ones(4096, device=DEV, dtype=BF), 0.5, True)
    assert y.dtype == BF and abs((y != 0).float().mean().item() - 0.5) < 0.05 and float(y.max()) == 2.0


@pytest.mark.parametrize("transA,transB", [(False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K1,K2", [(304, 200, 256, 0), (128, 128, 64, 0), (264, 136, 128, 192), (1024, 1024, 2048, 0),
                                       (472, 1024, 1024, 0), (1024, 480, 512, 0), (768, 512, 256, 128)])
def test_gemm_pipelined_transposed_operands_bit_equal_to_generic(ops, transA, transB, M, N, K1, K2):
    """dX / dW forms of the LDS-DMA kernel (k-major LDS images read with ds_read_b64_tr_b16) against the generic
    register-transposing kernel: bit-identical for every ring depth, incl. two-source K, split-K and f32 accumulation
    into C.  The two-wave-group variant (5; what the default policy 1 picks for small launches) sums even and odd K
    tiles separately: equal to rounding, and bitwise reproducible."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + K1)

    def operand(rows, K, tr):
        return torch.randn((K, rows) if tr else (rows, K), device=DEV, generator=g).to(BF)
    A1, B1 = operand(M, K1, transA), operand(N, K1, transB)
    A2, B2 = (operand(M, K2, transA), operand(N, K2, transB)) if K2 else (None, None)
    C0 = torch.randn(M, N, device=DEV, generator=g)
    outs = {}
    for pipe in (3, 7, 15, 8, 11, 16, 0, 5, 12, 1, 55):
        prev = lib.egk_gemm_set_pipeline(5 if pipe == 55 else pipe)
        try:
            out = C0.clone()
            ops.gemm(M, N, A1, A1.shape[1], B1, B1.shape[1], K1, out, N, A2=A2, lda2=A2.shape[1] if K2 else 0, B2=B2,
                     ldb2=B2.shape[1] if K2 else 0, K2=K2, transA=transA, transB=transB, accumulate=True)
            outs[pipe] = out
        finally:
            lib.egk_gemm_set_pipeline(prev)
    for v in (3, 7, 15, 8, 11, 16):  # one wave group (16: 192 x 128 tiles on 8 waves, row-major A): the MFMA chain of the
        assert torch.equal(outs[v], outs[0]), v  # generic kernel per accumulator, whatever the tile / ring depth
    assert torch.equal(outs[5], outs[55])
    # 64-row tiles with two wave groups (12; row-major A only): the same even / odd K sums as (5), whatever the tile height
    assert torch.equal(outs[12], outs[3] if transA else outs[5])
    assert torch.equal(outs[1], outs[5]) or torch.equal(outs[1], outs[3])
    torch.testing.assert_close(outs[5], outs[0], rtol=1e-5, atol=2e-3)
    opA = lambda t, tr: (t.t() if tr else t).double()
    ref = opA(A1, transA) @ opA(B1, transB).t() + (opA(A2, transA) @ opA(B2, transB).t() if K2 else 0) + C0.double()
    torch.testing.assert_close(outs[1], ref.float(), rtol=2e-3, atol=5e-3)


@pytest.mark.parametrize("M,N,K1,K2", [(300, 200, 256, 0), (128, 128, 64, 0), (257, 129, 128, 192), (2048, 1024, 1024, 0),
                                       (6144, 1024, 1024, 1024), (130, 478, 1024, 0), (16384, 1024, 512, 0), (12200, 1100, 128, 64),
                                       (768, 512, 256, 128)])
def test_gemm_pipelined_kernel_bit_equal_to_generic(ops, M, N, K1, K2):
    """The LDS-DMA pipelined kernel (bf16 row-major operands, K % 64 == 0) issues the same MFMA chain per
    accumulator as the generic kernel: results must be BIT-identical, for bf16 and f32 outputs, with the
    fused epilogue; and both must match an fp64 reference on the bf16-rounded operands."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + N + K1)
    A1 = torch.randn(M, K1, device=DEV, generator=g).to(BF)
    B1 = torch.randn(N, K1, device=DEV, generator=g).to(BF)
    A2 = torch.randn(M, K2, device=DEV, generator=g).to(BF) if K2 else None
    B2 = torch.randn(N, K2, device=DEV, generator=g).to(BF) if K2 else None
    bias = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g).to(BF)
    outs = {}
    for pipe in (3, 7, 15, 8, 11, 16, 0, 5, 12, 1):
        prev = lib.egk_gemm_set_pipeline(pipe)
        try:
            for dt in (BF, torch.float32):
                out = torch.empty(M, N, device=DEV, dtype=dt)
                ops.gemm(M, N, A1, K1, B1, K1, K1, out, N, A2=A2, lda2=K2, B2=B2, ldb2=K2, K2=K2, bias=bias, residual=res,
                         ldr=N, act=1)
                outs[(pipe, dt)] = out
        finally:
            lib.egk_gemm_set_pipeline(prev)
    for dt in (BF, torch.float32):
        assert torch.equal(outs[(3, dt)], outs[(0, dt)])
        assert torch.equal(outs[(8, dt)], outs[(0, dt)]) and torch.equal(outs[(11, dt)], outs[(0, dt)])  # 96- / 64-row tiles
        assert torch.equal(outs[(7, dt)], outs[(0, dt)])  # 256 x 256 tiles (the policy's choice for the two large outputs)
        assert torch.equal(outs[(15, dt)], outs[(0, dt)])  # 192 x 256 tiles (M % 192 == 0, N % 256 == 0: the (768, 512) case; 128 x 128 otherwise)
        assert torch.equal(outs[(16, dt)], outs[(0, dt)])  # 192 x 128 tiles, 8 waves, 3-stage ring (ragged M / N included)
        assert any(torch.equal(outs[(1, dt)], outs[(v, dt)]) for v in (5, 3, 8, 11, 7, 15, 16))
        assert torch.equal(outs[(12, dt)], outs[(5, dt)])  # 64-row tiles, two wave groups: the even / odd K sums of (5)
    # two wave groups: even / odd K tiles summed separately
    torch.testing.assert_close(outs[(5, torch.float32)], outs[(0, torch.float32)], rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(outs[(5, BF)].float(), outs[(0, BF)].float(), rtol=1e-2, atol=1e-2)
    if M * N <= 1 << 21:
        ref = A1.double() @ B1.double().t() + (A2.double() @ B2.double().t() if K2 else 0) + bias.double()
        ref = torch.relu(ref).float() + res.float()
        torch.testing.assert_close(outs[(1, torch.float32)], ref, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(outs[(5, torch.float32)], ref, rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("transA,transB", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K1,K2,splitk", [(300, 200, 256, 0, 1), (128, 128, 32, 0, 1), (264, 136, 128, 96, 1), (1024, 1024, 2048, 0, 4),
                                              (6144, 1024, 1024, 0, 1), (2048, 1024, 1024, 1024, 1), (472, 1024, 1024, 0, 2)])
def test_gemm_f32_pipelined_kernel_bit_equal_to_generic(ops, transA, transB, M, N, K1, K2, splitk):
    """The exact-f32 LDS-DMA kernel (f32 operands, K % 32 == 0: the kernel of the reference-precision mode) issues the k-ordered
    v_mfma_f32_16x16x4_f32 chain of the register-staged generic kernel per accumulator: BIT-identical results in all four
    operand layouts, with two K sources, split-K slabs, the fused epilogue (bias / ReLU / residual) and accumulation into C;
    and both match fp64 to f32 accumulation error."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + 3 * N + K1 + 7 * K2)

    def operand(rows, K, tr):
        return torch.randn((K, rows) if tr else (rows, K), device=DEV, generator=g)
    A1, B1 = operand(M, K1, transA), operand(N, K1, transB)
    A2, B2 = (operand(M, K2, transA), operand(N, K2, transB)) if K2 else (None, None)
    bias, res, C0 = torch.randn(N, device=DEV, generator=g), torch.randn(M, N, device=DEV, generator=g), torch.randn(M, N, device=DEV, generator=g)
    outs = {}
    for pipe in (1, 0):
        prev = lib.egk_gemm_set_pipeline(pipe)
        try:
            a = torch.empty(M, N, device=DEV)
            ops.gemm(M, N, A1, A1.shape[1], B1, B1.shape[1], K1, a, N, A2=A2, lda2=A2.shape[1] if K2 else 0, B2=B2,
                     ldb2=B2.shape[1] if K2 else 0, K2=K2, transA=transA, transB=transB, bias=bias, residual=res, ldr=N, act=1,
                     compute=ops.F32, splitk=splitk)
            b = C0.clone()
            ops.gemm(M, N, A1, A1.shape[1], B1, B1.shape[1], K1, b, N, A2=A2, lda2=A2.shape[1] if K2 else 0, B2=B2,
                     ldb2=B2.shape[1] if K2 else 0, K2=K2, transA=transA, transB=transB, accumulate=True, alpha=0.5,
                     compute=ops.F32, splitk=splitk)
            outs[pipe] = (a, b)
        finally:
            lib.egk_gemm_set_pipeline(prev)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    if M * N <= 1 << 21:
        opA = lambda t, tr: (t.t() if tr else t).double()
        dot = opA(A1, transA) @ opA(B1, transB).t() + (opA(A2, transA) @ opA(B2, transB).t() if K2 else 0)
        torch.testing.assert_close(outs[1][0], (torch.relu(dot + bias.double()) + res.double()).float(), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(outs[1][1], (0.5 * dot + C0.double()).float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(1024, 1024, 2048), (472, 1024, 1024), (128, 256, 4096), (1024, 4608, 6144), (115, 1024, 2048)])
def test_gemm_dw_with_fused_bias_gradient(ops, M, N, K):
    """dW launch with dbias: dbias[m] += sum_k dY[k, m], fused into the pipelined kernel (from the dY^T LDS image) or
    computed by the library's column-sum launches when the pipelined kernel is off -- both against fp64."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + N)
    ldg = (M + 7) // 8 * 8
    dY = torch.randn(K, ldg, device=DEV, generator=g).to(BF)[:, :M]  # padded row stride, as _operand_rows builds it
    X = torch.randn(K, N, device=DEV, generator=g).to(BF)
    W0, b0 = torch.randn(M, N, device=DEV, generator=g), torch.randn(M, device=DEV, generator=g)
    ref_w = (dY.double().t() @ X.double() + W0.double()).float()
    ref_b = (dY.double().sum(0) + b0.double()).float()
    res = {}
    for pipe in (3, 7, 0, 5, 1):  # (7 has no fused bias gradient: it must hand the launch to 3)
        prev = lib.egk_gemm_set_pipeline(pipe)
        try:
            w, b = W0.clone(), b0.clone()
            ops.gemm(M, N, dY, dY.stride(0), X, N, K, w, N, transA=True, transB=True, accumulate=True, dbias=b)
            res[pipe] = (w, b)
        finally:
            lib.egk_gemm_set_pipeline(prev)
    assert torch.equal(res[3][0], res[0][0]) and torch.equal(res[7][0], res[0][0])
    assert torch.equal(res[7][1], res[3][1])
    for pipe in (3, 7, 0, 5, 1):
        torch.testing.assert_close(res[pipe][0], ref_w, rtol=2e-3, atol=2e-2)
        torch.testing.assert_close(res[pipe][1], ref_b, rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("M,N,K,splitk", [(1024, 1024, 2048, 1), (472, 1024, 1024, 2), (1024, 4608, 6144, 1), (116, 1024, 2048, 4)])
def test_gemm_f32_dw_with_fused_bias_gradient(ops, M, N, K, splitk):
    """The exact-f32 dW launch with dbias: the pipelined f32 kernel sums its k-major dY image over k (as the bf16 kernel does)
    instead of a column-sum launch of its own; the weight gradient stays bit-identical to the generic kernel's, the bias
    gradient agrees with fp64 to f32 summation error; split-K slabs included."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    dY = torch.randn(K, M, device=DEV, generator=g)
    X = torch.randn(K, N, device=DEV, generator=g)
    W0, b0 = torch.randn(M, N, device=DEV, generator=g), torch.randn(M, device=DEV, generator=g)
    res = {}
    for pipe in (1, 0):
        prev = lib.egk_gemm_set_pipeline(pipe)
        try:
            w, b = W0.clone(), b0.clone()
            ops.gemm(M, N, dY, M, X, N, K, w, N, transA=True, transB=True, accumulate=True, dbias=b, compute=ops.F32, splitk=splitk)
            res[pipe] = (w, b)
        finally:
            lib.egk_gemm_set_pipeline(prev)
    assert torch.equal(res[1][0], res[0][0])
    ref_b = (dY.double().sum(0) + b0.double()).float()
    for pipe in (1, 0):
        torch.testing.assert_close(res[pipe][1], ref_b, rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(res[1][0], (dY.double().t() @ X.double() + W0.double()).float(), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("layout", ["tt", "nn", "nt"])
def test_gemm_f32_grouped_launch_equals_the_single_launches(ops, layout):
    """Up to eight exact-f32 contractions of one layout in ONE launch (the weight gradients of the reference-precision step:
    no split-K slabs, no reduce launches): every output BIT-identical to the problem's own launch, the fused bias gradients
    of the dW form to f32 summation error."""
    g = torch.Generator(device=DEV).manual_seed(len(layout) + 11)
    tA, tB = layout[0] == "t", layout[1] == "t"
    shapes = [(1024, 1024, 2048), (1024, 1024, 2048), (472, 1024, 1024), (1024, 4608, 2048), (128, 256, 4096), (115 if not tA else 116, 1024, 2048)]
    probs, singles = [], []
    for M, N, K in shapes:
        A = torch.randn((K, M) if tA else (M, K), device=DEV, generator=g)
        B = torch.randn((K, N) if tB else (N, K), device=DEV, generator=g)
        C0 = torch.randn(M, N, device=DEV, generator=g)
        b0 = torch.randn(M, device=DEV, generator=g) if layout == "tt" else None
        cg, cs = C0.clone(), C0.clone()
        bg, bs = (b0.clone(), b0.clone()) if b0 is not None else (None, None)
        kw = dict(transA=tA, transB=tB, accumulate=True, compute=ops.F32)
        probs.append(((M, N, A, A.shape[1], B, B.shape[1], K, cg, N), dict(kw, dbias=bg) if bg is not None else kw))
        ops.gemm(M, N, A, A.shape[1], B, B.shape[1], K, cs, N, allow_splitk=False, **(dict(kw, dbias=bs) if bs is not None else kw))
        singles.append((cs, bs, cg, bg, A, b0))
    ops.gemm_grouped(probs)
    torch.cuda.synchronize()
    for cs, bs, cg, bg, A, b0 in singles:
        assert torch.equal(cs, cg)
        if bs is not None:
            torch.testing.assert_close(bg, (A.double().sum(0) + b0.double()).float(), rtol=1e-5, atol=1e-3)
            torch.testing.assert_close(bg, bs, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("n_out", [478, 115, 2])
def test_classifier_dx_on_zero_padded_k_axis(ops, n_out):
    """A classifier layer (rows of W not a multiple of 64) whose parameters live in FlatAdam's flat buffers: the
    bf16 copy of W is kept zero-padded to whole 64-row blocks and the loss gradient is converted with zero columns,
    so dX = dY @ W contracts over a padded K on the pipelined kernel.  dX, dW, db against fp64 on the bf16-rounded
    operands, before and after an optimizer step (the padding must stay zero under Adam)."""
    from egopack_amd import _lib
    from egopack_amd.optim import FlatAdam
    g = torch.Generator(device=DEV).manual_seed(n_out)
    M, K = 2048, 1024
    W = (torch.randn(n_out, K, device=DEV, generator=g) / 32).requires_grad_(True)
    b = torch.randn(n_out, device=DEV, generator=g).requires_grad_(True)
    opt = FlatAdam([W, b], lr=1e-3, weight_decay=1e-5)
    lib = _lib.load()
    for it in range(3):
        x = torch.randn(M, K, device=DEV, generator=g).to(BF).requires_grad_(True)
        wt = torch.randn(M, n_out, device=DEV, generator=g)
        opt.zero_grad()
        with ops.compute_mode("bf16"):
            y = ops.linear(x, W, b, out_f32=True)
            lib.egk_prof_reset()
            ops.prof_enable(True)
            (y * wt).sum().backward()
            ops.prof_enable(False)
        if it > 0:  # flat buffers exist: the padded copy is in use and the dX launch is a pipelined NT contraction
            assert W._egk_shadow_rows64.shape == ((n_out + 63) // 64 * 64, K)
            assert torch.count_nonzero(W._egk_shadow_rows64[n_out:]) == 0
        Wr, xr, gr = W.detach().to(BF).double(), x.detach().double(), wt.to(BF).double()
        torch.testing.assert_close(x.grad.float(), (gr @ Wr).float(), rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(W.grad, (gr.t() @ xr).float(), rtol=2e-3, atol=5e-2)
        torch.testing.assert_close(b.grad, gr.sum(0).float(), rtol=2e-3, atol=5e-2)
        torch.testing.assert_close(y, (xr @ Wr.t() + b.detach().double()).float(), rtol=2e-3, atol=2e-2)
        opt.step()


@pytest.mark.parametrize("rows,K,k", [(7, 5, 3), (33, 257, 4), (64, 4096, 4), (17, 1000, 16), (5, 64, 1), (9, 130, 9)])
def test_topk_selection_equals_full_lexicographic_sort(ops, rows, K, k):
    """egk_topk_smallest on a given dot-product matrix: the k picks of every row must equal a full (distance, index)
    sort of d = 1 - dot * f_inv * b_inv (same f32 expression) -- with many exact ties, +/-inf and NaN distances."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(rows * 1000 + K)
    dot = (torch.randint(-3, 4, (rows, K), generator=g).float() / 4)  # few distinct values: ties everywhere
    dot[0, :] = 0.5  # a row of identical distances -> indices 0..k-1
    if K > 8:
        dot[1, 3] = float("nan")
        dot[1, 5] = float("inf")   # distance -inf: the nearest
        dot[2, 1] = float("-inf")  # distance +inf: the farthest
    # scales that are powers of two: every product and difference below is exact in f32, so the expected order does
    # not depend on whether the device contracts 1 - a*b into an FMA
    f_inv = 2.0 ** torch.randint(-1, 2, (rows,), generator=g).float()
    b_inv = 2.0 ** torch.randint(-1, 2, (K,), generator=g).float()
    d = 1.0 - dot * f_inv[:, None] * b_inv[None, :]
    key = torch.where(torch.isnan(d), torch.full_like(d, float("inf")), d)
    # NaN never wins against a number: order them after everything else; ties by index (stable sort)
    order = torch.sort(key + torch.where(torch.isnan(d), torch.tensor(float("inf")), torch.tensor(0.0)), dim=1, stable=True).indices
    nn = torch.empty((rows, k), dtype=torch.int64, device=DEV)
    d_dot, d_f, d_b = dot.to(DEV), f_inv.to(DEV), b_inv.to(DEV)  # keep the device copies alive across the raw call
    rc = lib.egk_topk_smallest(ops._stream(), ops._p(d_dot), K, ops._p(d_f), ops._p(d_b), ops._p(nn), rows, K, k)
    assert rc == 0
    nn = nn.cpu()
    valid = (~torch.isnan(d)).sum(1)
    for r in range(rows):
        kk = min(k, int(valid[r]))
        assert nn[r, :kk].tolist() == order[r, :kk].tolist(), r


@pytest.mark.parametrize("dtype", [torch.float32, BF])
@pytest.mark.parametrize("T,B", [(256, 3), (70, 5), (20, 2), (32, 6), (66, 2), (67, 2)])
def test_csr_gather_rows_with_hundreds_of_edges(ops, dtype, T, B):
    """LTA connectivity at long T: the fan-out node has out-degree T - 1 (backward orientation) -- listed by
    build_csr and cut over several workgroups by the kernel; both orientations and the gated transposed form against
    fp64, incl. a fan-IN node (forward orientation) and rows just below / above the threshold."""
    from egopack_amd import data as D
    g = torch.Generator().manual_seed(T + B)
    eis, off = [], 0
    for b in range(B):
        y = torch.ones(T, 2, dtype=torch.long)
        y[:2] = -1
        ei = D.lta_connectivity_edges(torch.arange(T), y, 1.5)
        fan_in = torch.stack([torch.arange(3, T), torch.full((T - 3,), 2)])  # every later node -> node 2
        eis.append(torch.cat([ei, fan_in], 1) + off)
        off += T
    N, H = off, 256
    graph = D.build_csr(torch.cat(eis, 1), N)
    if T - 4 > D.HEAVY_DEGREE:
        assert graph.t_heavy.numel() >= B and graph.heavy.numel() >= B
    else:
        assert graph.t_heavy.numel() == 0 and graph.heavy.numel() == 0  # below the threshold: the in-workgroup path
    gd = graph.to(DEV)
    x = torch.randn(N, H, generator=g).to(dtype)
    gate = torch.randn(N, H, generator=g).to(dtype)
    xd, gated = x.to(DEV), gate.to(DEV)
    A = torch.zeros(N, N, dtype=torch.float64)
    src, tgt = torch.cat(eis, 1)
    A.index_put_((tgt, src), torch.ones(src.numel(), dtype=torch.float64), accumulate=True)
    deg = A.sum(1).clamp(min=1)
    ref_f = (A / deg[:, None]) @ x.double()
    ref_b = ((A / deg[:, None]).t() @ x.double()) * (gate.double() > 0)
    out = torch.empty_like(xd)
    ops._csr_gather(xd, gd.rowptr, gd.col, None, None, out, gd.heavy)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(out.float().cpu().double(), ref_f, **tol)
    out2 = torch.empty_like(xd)
    ops._csr_gather(xd, gd.t_rowptr, gd.t_col, gd.t_wgt, gated, out2, gd.t_heavy)
    torch.testing.assert_close(out2.float().cpu().double(), ref_b, **tol)
    out3 = torch.empty_like(xd)  # reproducible bit for bit
    ops._csr_gather(xd, gd.t_rowptr, gd.t_col, gd.t_wgt, gated, out3, gd.t_heavy)
    assert torch.equal(out2, out3)
    # heavy_mode 1 (listed rows of at most HEAVY_IN_LAUNCH_DEGREE edges: one workgroup each inside the launch) is what
    # build_csr picks for them, and both modes are legal for any list: same sums (mode 1 in edge order, mode 0 in chunks)
    max_out = int((graph.t_rowptr[1:] - graph.t_rowptr[:-1]).max())
    assert graph.t_heavy_mode == int(graph.t_heavy.numel() > 0 and max_out <= D.HEAVY_IN_LAUNCH_DEGREE)
    for mode in (0, 1):
        o_f, o_b = torch.empty_like(xd), torch.empty_like(xd)
        ops._csr_gather(xd, gd.rowptr, gd.col, None, None, o_f, gd.heavy, mode)
        ops._csr_gather(xd, gd.t_rowptr, gd.t_col, gd.t_wgt, gated, o_b, gd.t_heavy, mode)
        torch.testing.assert_close(o_f.float().cpu().double(), ref_f, **tol)
        torch.testing.assert_close(o_b.float().cpu().double(), ref_b, **tol)


# ---------------------------------------------------------------------------------------------------------
# grouped launches (the projection heads of several task batches as one chain)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("layout", ["nn", "nt", "tt"])
@pytest.mark.parametrize("sizes", [[(2048, 1024, 1024)] * 3, [(256, 128, 64), (192, 256, 64), (64, 128, 128), (448, 128, 64)],
                                   [(6144, 1024, 1024), (2048, 1024, 1024)],
                                   [(64, 1024, 1024), (2048, 1024, 1024), (2048, 1024, 1024)]])  # (the heads' group: 192-row tiles + loaders)
def test_gemm_grouped_equals_separate_contractions(ops, layout, sizes):
    """egk_gemm_grouped (one launch, blockIdx.y = problem) against the fp64 product of the bf16-rounded operands, and
    against egk_gemm problem by problem (same kernel body; the tile variant may differ: fp32 accumulation-order noise only)."""
    g = gen(len(sizes) * 7 + len(layout))
    ta, tb = layout[0] == "t", layout[1] == "t"
    probs, refs, outs, singles = [], [], [], []
    for (M, N, K) in sizes:
        if layout == "tt":
            K = (K + 63) // 64 * 64
        A = torch.randn((K, M) if ta else (M, K), generator=g).to(torch.bfloat16)
        B = torch.randn((K, N) if tb else (N, K), generator=g).to(torch.bfloat16)
        bias = torch.randn(N, generator=g)
        opA = (A.t() if ta else A).double()
        opB = (B.t() if tb else B).double()
        acc = layout == "tt"
        c0 = torch.randn(M, N, generator=g) if acc else torch.zeros(M, N)
        refs.append(opA @ opB.t() + (c0.double() if acc else bias.double()))
        Ad, Bd = A.to(DEV), B.to(DEV)
        out = c0.clone().to(DEV) if acc else torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        one = out.clone()
        kw = dict(transA=ta, transB=tb, compute=ops.BF16)
        kw.update(dict(accumulate=True) if acc else dict(bias=bias.to(DEV)))
        args = (M, N, Ad, Ad.stride(0), Bd, Bd.stride(0), K, out, N)
        probs.append((args, kw))
        outs.append(out)
        ops.gemm(M, N, Ad, Ad.stride(0), Bd, Bd.stride(0), K, one, N, allow_splitk=False, **kw)
        singles.append(one)
    ops.gemm_grouped(probs)
    for out, one, ref in zip(outs, singles, refs):
        tol = dict(rtol=1e-3, atol=1e-3) if out.dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
        torch.testing.assert_close(out.double().cpu(), ref, **tol)
        # (bf16 results: another tile variant = another f32 accumulation order = now and then the neighbouring bf16 value)
        torch.testing.assert_close(out.float(), one.float(), rtol=1e-5 if out.dtype == torch.float32 else 8e-3,
                                   atol=1e-4 if out.dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("with_dbias", [False, True])
def test_gemm_grouped_tall_tiles_for_uneven_weight_gradient_groups_bit_equal(ops, with_dbias):
    """A weight-gradient group whose 128 x 128 tiling leaves the second workgroup slot of many CUs empty (the temporal pooling's three
    weight gradients: 288 + 64 + 64 tiles) runs on 256 x 128 tiles (8 waves, 3-stage ring; egk_gemm_set_pipeline(850 / 851) off / on):
    the same accumulation order per output element -- bit for bit the 128 x 128 launch, fused bias gradient included."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(77)
    sizes = [(1024, 4608, 4096), (1024, 1024, 4096), (1024, 1024, 4096)]  # (K walks of >= 64 tiles: the policy's range)
    outs = {}
    for knob in (850, 851):
        lib.egk_gemm_set_pipeline(knob)
        try:
            g.manual_seed(77)
            probs, res = [], []
            for (M, N, K) in sizes:
                A = torch.randn(K, M, generator=g).to(torch.bfloat16).to(DEV)  # dY [rows, M]
                B = torch.randn(K, N, generator=g).to(torch.bfloat16).to(DEV)  # X  [rows, N]
                c = torch.randn(M, N, generator=g).to(DEV)
                b = torch.randn(M, generator=g).to(DEV)
                kw = dict(transA=True, transB=True, compute=ops.BF16, accumulate=True)
                if with_dbias:
                    kw["dbias"] = b
                probs.append(((M, N, A, A.stride(0), B, B.stride(0), K, c, N), kw))
                res.append((c, b))
            ops.gemm_grouped(probs)
            torch.cuda.synchronize()
            outs[knob] = [(c.clone(), b.clone()) for c, b in res]
        finally:
            lib.egk_gemm_set_pipeline(851)
    for (c0, b0), (c1, b1), (M, N, K) in zip(outs[850], outs[851], sizes):
        assert torch.equal(c0, c1), (M, N, K)
        assert torch.equal(b0, b1), (M, N, K)
    # and the values: against the fp64 product for the first problem
    g.manual_seed(77)
    M, N, K = sizes[0]
    A = torch.randn(K, M, generator=g).to(torch.bfloat16)
    B = torch.randn(K, N, generator=g).to(torch.bfloat16)
    c = torch.randn(M, N, generator=g)
    torch.testing.assert_close(outs[851][0][0].double().cpu(), A.double().t() @ B.double() + c.double(), rtol=1e-3, atol=2e-2)


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(6144, 1024, 1024), (6000, 1024, 2048), (16384, 1024, 1024), (6144, 2048, 256), (6144, 1024, 4608)])
def test_gemm_192_row_tile_with_loader_waves_bit_equal(ops, tb, M, N, K):
    """The 192 x 128 tile with four LOADER waves (they issue every LDS-DMA piece of the workgroup; the eight compute waves only read
    fragments and feed the matrix pipe; egk_gemm_set_pipeline(870 / 871) off / on) against the same tile whose compute waves issue
    their own pieces: the same accumulation order per element -- bit for bit, bias + activation epilogue included, ragged row counts
    and several rounds of tiles included."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(M + N + K + int(tb))
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
    B = torch.randn((K, N) if tb else (N, K), generator=g).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    outs = {}
    prev = lib.egk_gemm_set_pipeline(16)
    try:
        for knob in (870, 871):
            lib.egk_gemm_set_pipeline(knob)
            out = torch.zeros(M, N, device=DEV, dtype=torch.bfloat16)
            ops.gemm(M, N, A, K, B, B.shape[1], K, out, N, transB=tb, bias=bias, act=1, compute=ops.BF16, allow_splitk=False)
            torch.cuda.synchronize()
            outs[knob] = out
    finally:
        lib.egk_gemm_set_pipeline(871)
        lib.egk_gemm_set_pipeline(prev)
    assert torch.equal(outs[870], outs[871])
    ref = torch.relu(A.double().cpu() @ (B.double().cpu() if tb else B.double().cpu().t()) + bias.double().cpu())
    torch.testing.assert_close(outs[871].double().cpu(), ref, rtol=2e-2, atol=2e-1)


def test_gemm_grouped_rejects_what_it_cannot_run(ops):
    A = torch.randn(64, 96, device=DEV).to(torch.bfloat16)  # K = 96 is not a multiple of 64
    B = torch.randn(64, 96, device=DEV).to(torch.bfloat16)
    out = torch.empty(64, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        ops.gemm_grouped([((64, 64, A, 96, B, 96, 96, out, 64), {}), ((64, 64, A, 96, B, 96, 96, out, 64), {})])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("cols,rows", [(1024, [2048, 2048, 2048]), (256, [64, 192, 5]), (640, [100, 0, 37, 64])])
def test_rowln_grouped_equals_per_range_launches(ops, dt, cols, rows):
    """The grouped row LayerNorm + ReLU (one launch, one parameter pair per row range) forward and backward against the
    single-range kernel run range by range: y / dx to last-bit noise (a few % of the elements differ in the last bit), dw / db to summation-order noise."""
    import ctypes as C
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(cols + len(rows))
    G, n = len(rows), sum(rows)
    ptr = [0]
    for r in rows:
        ptr.append(ptr[-1] + r)
    x = torch.randn(n, cols, generator=g).to(dt).to(DEV)
    dy = torch.randn(n, cols, generator=g).to(dt).to(DEV)
    ws_, bs_ = [torch.randn(cols, generator=g).to(DEV) for _ in rows], [torch.randn(cols, generator=g).to(DEV) for _ in rows]
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    row_ptr = (C.c_int32 * (G + 1))(*ptr)
    s = ops._stream()
    assert lib.egk_rowln_group_fwd(s, ops._p(x), ops._ptr_array(ws_), ops._ptr_array(bs_), row_ptr, G, ops._p(y), ops._p(mean),
                                   ops._p(rstd), cols, 1e-5, 1, ops._dt(x)) == 0
    grid = lib.egk_rowln_bwd_ws_rows(max(rows))
    ws = torch.zeros(G * grid * 2 * cols, device=DEV)
    assert lib.egk_rowln_group_bwd(s, ops._p(dy), ops._p(x), ops._ptr_array(ws_), ops._ptr_array(bs_), row_ptr, G, ops._p(mean),
                                   ops._p(rstd), ops._p(dx), ops._p(ws), cols, 1, ops._dt(x)) == 0
    for k in range(G):
        if rows[k] == 0:
            continue
        xs = x[ptr[k]:ptr[k + 1]].clone().requires_grad_(True)
        w1, b1 = ws_[k].clone().requires_grad_(True), bs_[k].clone().requires_grad_(True)
        y1 = ops.row_layernorm(xs, w1, b1, 1e-5, relu=True)
        y1.backward(dy[ptr[k]:ptr[k + 1]])
        # (same source, two kernel instantiations: the compiler may contract a multiply-add differently -> last-bit noise)
        ulp = dict(rtol=2e-6, atol=2e-6) if dt == torch.float32 else dict(rtol=8e-3, atol=8e-3)
        torch.testing.assert_close(y[ptr[k]:ptr[k + 1]].float(), y1.detach().float(), **ulp)
        torch.testing.assert_close(dx[ptr[k]:ptr[k + 1]].float(), xs.grad.float(), **ulp)
        dw, db = torch.zeros(cols, device=DEV), torch.zeros(cols, device=DEV)
        part = ws[k * grid * 2 * cols:]
        assert lib.egk_ln_bwd_reduce(s, ops._p(part), ops._p(dw), ops._p(db), max(rows), cols, 0) == 0
        tol = dict(rtol=1e-4, atol=1e-4 * max(1.0, rows[k] ** 0.5))
        torch.testing.assert_close(dw, w1.grad, **tol)
        torch.testing.assert_close(db, b1.grad, **tol)


# ---------------------------------------------------------------------------------------------------------
# graph-LayerNorm statistics taken in the epilogue of the contraction that produces its input / its output gradient
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,segs", [(6144, [0, 2048, 4096, 6144]), (2048, [0, 2048]), (1000, [0, 300, 1000]), (4500, [0, 1000, 2500, 4500])])
def test_gemm_epilogue_segment_statistics_feed_graph_layernorm(ops, M, segs):
    """egk_gemm st_mode 1 / 2: the per-tile partials of (sum, sum^2) of the result, resp. of the LayerNorm-backward sums,
    summed over tiles equal the direct fp64 sums of the STORED (bf16-rounded) result; the apply-only LayerNorm entry points
    fed with them reproduce the two-pass kernels (y, dx, dw, db), tiles straddling segment boundaries included."""
    import ctypes as C
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(M)
    H = 1024 if M > 1500 else 256
    K = 512
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W = (torch.randn(H, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    bias = torch.randn(H, generator=g).to(DEV)
    seg = torch.tensor(segs, dtype=torch.int32, device=DEV)
    n_seg, min_rows = len(segs) - 1, min(b - a for a, b in zip(segs, segs[1:]))
    lw, lb = (torch.randn(H, generator=g) * 0.5 + 1).to(DEV), (torch.randn(H, generator=g) * 0.2).to(DEV)
    out = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    args, kw = (M, H, A, K, W, K, K, out, H), dict(bias=bias, compute=ops.BF16)
    got = ops._gemm_with_stats(args, kw, dict(mode=1, seg_ptr=seg, n_seg=n_seg, min_rows=min_rows))
    if min_rows < 64:
        assert got is None
        return
    assert got is not None, "the rows-epilogue variants must take the statistics at these shapes"
    ws, blocks = got
    part = ws.view(blocks, n_seg, 2).sum(0).cpu()
    ref_out = torch.empty_like(out)
    ops.gemm(*args[:7], ref_out, H, **kw)
    assert torch.equal(out, ref_out)  # the statistics do not touch the result
    o64 = out.double().cpu()
    for s_ in range(n_seg):
        blk = o64[segs[s_]:segs[s_ + 1]]
        torch.testing.assert_close(part[s_, 0], blk.sum(), rtol=1e-6, atol=1e-3)
        torch.testing.assert_close(part[s_, 1], (blk * blk).sum(), rtol=1e-6, atol=1e-3)
    # forward: apply-only entry point == the two-pass kernel
    y_ref = ops.graph_layernorm_lrelu(out, lw, lb, seg)
    y, lnctx = ops.graph_layernorm_lrelu(out, lw, lb, seg, partials=(ws, blocks), min_seg_rows=min_rows, return_ctx=True)
    torch.testing.assert_close(y.float(), y_ref.float(), rtol=1e-2, atol=1e-2)
    assert (y != y_ref).float().mean() < 0.02  # (last-bit differences of the segment mean / std only)
    # backward: dy produced by a dX-shaped contraction with st_mode 2
    G = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W2 = (torch.randn(K, H, generator=g) * 0.1).to(torch.bfloat16).to(DEV)  # [K, H]: transB layout, dy = G @ W2
    dy = torch.empty(M, H, dtype=torch.bfloat16, device=DEV)
    ops._ln_bwd_stats_launch(lnctx, (M, H, G, K, W2, H, K, dy, H), dict(transB=True, compute=ops.BF16), dy)
    assert lnctx["bwd"] is not None and lnctx["bwd"][2] == dy.data_ptr()
    ws2, blocks2, _ = lnctx["bwd"]
    x64, dy64 = out.double().cpu(), dy.double().cpu()
    stats = lnctx["stats"].double().cpu().view(n_seg, 2)
    part2 = ws2.view(blocks2, n_seg, 2).sum(0).cpu()
    for s_ in range(n_seg):
        xs, ds = x64[segs[s_]:segs[s_ + 1]], dy64[segs[s_]:segs[s_ + 1]]
        xh = (xs - stats[s_, 0]) * stats[s_, 1]
        pre = xh * lw.double().cpu() + lb.double().cpu()
        dxh = ds * torch.where(pre > 0, 1.0, 0.2) * lw.double().cpu()
        torch.testing.assert_close(part2[s_, 0], dxh.sum(), rtol=2e-4, atol=5e-2)
        torch.testing.assert_close(part2[s_, 1], (dxh * xh).sum(), rtol=2e-4, atol=5e-2)
    # autograd through both paths
    def grads(use_pre):
        xin = out.clone().requires_grad_(True)
        w1, b1 = lw.clone().requires_grad_(True), lb.clone().requires_grad_(True)
        yy, ctx2 = ops.graph_layernorm_lrelu(xin, w1, b1, seg, min_seg_rows=min_rows, return_ctx=True)
        if use_pre:
            ops._ln_bwd_stats_launch(ctx2, (M, H, G, K, W2, H, K, dy, H), dict(transB=True, compute=ops.BF16), dy)
        yy.backward(dy)
        return xin.grad, w1.grad, b1.grad
    (dx1, dw1, db1), (dx0, dw0, db0) = grads(True), grads(False)
    torch.testing.assert_close(dx1.float(), dx0.float(), rtol=2e-2, atol=2e-3)
    torch.testing.assert_close(dw1, dw0, rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(db1, db0, rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pe_add_table_is_bit_identical_to_the_direct_evaluation(ops, dt):
    """x + PE(pos) from the cached per-position table (egk_pe_table + egk_pe_add_table) against the per-node evaluation
    (egk_pe_add): the same sinf / cosf values, so the same bits; positions outside the announced range fall back to the
    direct evaluation inside the kernel."""
    g = gen(17)
    rows, cols = 300, 1024
    x = torch.randn(rows, cols, generator=g).to(dt).to(DEV)
    pos = torch.randint(-16, 16, (rows,), generator=g).to(DEV)
    freq = torch.logspace(0, 1, cols // 2, 1e-4).to(DEV)
    ref = ops.pe_add(x, pos, freq)
    assert torch.equal(ops.pe_add(x, pos, freq, (-16, 15)), ref)
    assert torch.equal(ops.pe_add(x, pos, freq, (-4, 3)), ref)  # most positions outside the table
    torch.testing.assert_close(ref.float().cpu(), (x.float().cpu() + P.positional_encoding(pos.cpu(), freq.cpu())).to(dt).float(),
                               rtol=1e-2 if dt == torch.bfloat16 else 1e-5, atol=1e-2 if dt == torch.bfloat16 else 1e-5)


def _adam_case(n, seed):
    g = gen(seed)
    p, gr = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    m, v = (0.1 * torch.randn(n, generator=g)).to(DEV), (0.01 * torch.rand(n, generator=g)).to(DEV)
    hyper = torch.tensor([1e-3, 1.0 - 0.9 ** 3, (1.0 - 0.999 ** 3) ** 0.5, 0.5], dtype=torch.float32, device=DEV)
    return p, gr, m, v, hyper


def test_zero_fill_over_ranges_clears_the_ranges_and_nothing_else(ops):
    """egk_zero_fill_ranges (the step's gradient clear once the single-writer slots are left out; optimizer.zero_grad(), reference
    main_temporal.py:76): whole 16-byte groups of up to 48 ranges in one launch, everything else untouched; ragged or too many
    ranges are refused."""
    import ctypes as C
    from egopack_amd import _lib
    lib = _lib.load()
    buf = torch.arange(1, 100001, dtype=torch.float32, device=DEV)
    want = buf.clone()
    ranges = [(0, 16), (64, 4000), (8192, 0), (20000, 80000 - 20000), (99996, 4)]
    for b, n in ranges:
        want[b:b + n] = 0
    bg, ln = (C.c_int64 * len(ranges))(*[4 * b for b, _ in ranges]), (C.c_int64 * len(ranges))(*[4 * n for _, n in ranges])
    assert lib.egk_zero_fill_ranges(ops._stream(), ops._p(buf), bg, ln, len(ranges)) == 0, _lib.last_error()
    assert torch.equal(buf, want)
    odd = (C.c_int64 * 1)(8), (C.c_int64 * 1)(16)
    assert lib.egk_zero_fill_ranges(ops._stream(), ops._p(buf), odd[0], odd[1], 1) != 0 and "16-byte" in _lib.last_error()
    many = (C.c_int64 * 49)(*([0] * 49)), (C.c_int64 * 49)(*([16] * 49))
    assert lib.egk_zero_fill_ranges(ops._stream(), ops._p(buf), many[0], many[1], 49) != 0
    assert torch.equal(buf, want)


def test_adam_step_constants_are_computed_on_the_device_and_follow_lr_and_the_step_count(ops):
    """egk_adam_hyper: hyper = {lr, 1 - b1^t, sqrt(1 - b2^t), grad_scale} from a device-side step counter -- the host's
    double arithmetic rounded to f32 once -- so that a captured step carries the launch and a replay needs no host -> device
    copy.  The source is uploaded only when lr / grad_scale change, the counter only when ``step_count`` is set from outside."""
    import math
    from egopack_amd.optim import FlatAdam
    p = torch.nn.Parameter(torch.randn(1000, device=DEV))
    p.grad = torch.randn(1000, device=DEV)
    opt = FlatAdam([p], lr=3e-4, betas=(0.9, 0.999))

    def want(t, lr, gs):
        return torch.tensor([lr, 1.0 - 0.9 ** t, math.sqrt(1.0 - 0.999 ** t), gs], dtype=torch.float32)
    for t in range(1, 6):
        opt.step()
        assert torch.equal(opt._hyper.cpu(), want(t, 3e-4, 1.0)), t
    assert opt.step_count == 5 and int(opt._t_dev.item()) == 5 and opt._t_mirror == 5
    opt.param_groups[0]["lr"] = 1e-5  # an lr schedule stepping
    opt.grad_scale = 0.125            # a gradient exchange over 8 ranks
    opt.step()
    assert torch.equal(opt._hyper.cpu(), want(6, 1e-5, 0.125))
    opt.step_count = 123456           # a checkpoint being resumed
    opt.step()
    assert torch.equal(opt._hyper.cpu(), want(123457, 1e-5, 0.125)) and int(opt._t_dev.item()) == 123457
    # captured: the launch inside a graph advances the counter with every replay
    g = torch.cuda.CUDAGraph()
    opt.sync_hyper_source()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            opt.prepare_hyper(in_capture=True)
            opt.launch()
    torch.cuda.current_stream().wait_stream(side)
    for k in range(3):
        opt.sync_hyper_source()
        g.replay()
        opt.note_captured_step()
        opt.step_count += 1
        assert torch.equal(opt._hyper.cpu(), want(123458 + k, 1e-5, 0.125))
    assert int(opt._t_dev.item()) == opt.step_count == opt._t_mirror == 123460


# ---------------------------------------------------------------------------------------------------------
# round 4: the bf16 x 1024-column row kernels (csrc/rows1024.h) against the generic ones and against fp64
# ---------------------------------------------------------------------------------------------------------
class _rows_v2:
    """``with _rows_v2(False):`` -- the generic row kernels (egk_tune 3)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        from egopack_amd import _lib
        self.prev = _lib.load().egk_tune(3, 1 if self.on else 0)

    def __exit__(self, *a):
        from egopack_amd import _lib
        _lib.load().egk_tune(3, self.prev)


def test_rows1024_gather_and_positional_encoding_are_bit_identical_to_the_generic_kernels(ops):
    """Banded mean gather (coded rows and general rows: LTA forecast nodes, isolated rows) and x + PE(pos) from the table on bf16
    [N, 1024]: the same sums in the same order -- bit for bit."""
    from egopack_amd import data as D
    g = gen(4242)
    parts, n = [], 0
    for T in (32, 9, 1, 32, 5):
        parts.append(D.radius_band_edges(torch.arange(T), 1) + n)
        n += T
    y = torch.stack([torch.randint(1, 5, (14,), generator=g), torch.randint(0, 5, (14,), generator=g)], 1)
    y[:3] = -1
    parts.append(D.lta_connectivity_edges(torch.arange(14), y, 1.5) + n)
    n += 14 + 2  # + isolated rows
    for rep in range(40):  # enough rows for several sweeps per wave
        parts.append(D.radius_band_edges(torch.arange(32), 1) + n)
        n += 32
    ei = torch.cat(parts, 1)
    graph = D.build_csr(ei, n).to(DEV)
    assert (graph.band == 0xFF).any() and (graph.band == 5).any() and (graph.band == 0).any()
    x = torch.randn(n, 1024, generator=g).to(DEV).to(BF)
    outs = []
    for v2 in (True, False):
        with _rows_v2(v2):
            o = torch.empty_like(x)
            ops._csr_gather(x, graph.rowptr, graph.col, None, None, o, graph.heavy, graph.heavy_mode, band=graph.band)
            outs.append(o)
    assert torch.equal(outs[0], outs[1])
    ref = P.scatter_mean(x.float().cpu()[ei[0]], ei[1], n)
    torch.testing.assert_close(outs[0].float().cpu(), ref, **OUT16)
    # listed rows beyond the in-launch limit (a 256-node LTA sequence whose forecast nodes see every input node): the banded
    # kernel skips them, the split launches produce them -- again the generic kernel's result, bit for bit
    T2 = 256
    y2 = torch.stack([torch.randint(1, 5, (T2,), generator=g), torch.randint(0, 5, (T2,), generator=g)], 1)
    y2[:T2 - 100] = -1
    ei2 = torch.cat([D.radius_band_edges(torch.arange(T2), 1), D.lta_connectivity_edges(torch.arange(T2), y2, float(T2)) + T2,
                     D.radius_band_edges(torch.arange(T2), 1) + 2 * T2], 1)
    g2 = D.build_csr(ei2, 3 * T2).to(DEV)
    assert g2.heavy is not None and g2.heavy.numel() > 0 and g2.heavy_mode == 0
    x2 = torch.randn(3 * T2, 1024, generator=g).to(DEV).to(BF)
    o2 = []
    for v2 in (True, False):
        with _rows_v2(v2):
            o = torch.full_like(x2, 3.0)
            ops._csr_gather(x2, g2.rowptr, g2.col, None, None, o, g2.heavy, g2.heavy_mode, band=g2.band)
            o2.append(o)
    assert torch.equal(o2[0], o2[1])
    torch.testing.assert_close(o2[0].float().cpu(), P.scatter_mean(x2.float().cpu()[ei2[0]], ei2[1], 3 * T2), **OUT16)
    pos = (torch.arange(n) % 32 - 16).to(DEV)
    freq = torch.logspace(0, 1, 512, 1e-4).to(DEV)
    pes = []
    for v2 in (True, False):
        with _rows_v2(v2):
            pes.append(ops.pe_add(x, pos, freq, (-16, 15)))
            pes.append(ops.pe_add(x, pos, freq, (-4, 3)))  # most positions outside the table: evaluated directly
    assert torch.equal(pes[0], pes[2]) and torch.equal(pes[1], pes[3]) and torch.equal(pes[0], pes[1])


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("weighted,gated", [(True, True), (True, False), (False, True), (False, False)])
def test_rows1024_general_gather_at_96_registers_against_the_generic_kernel(ops, weighted, gated, split):
    """csr_gather_1k_kernel (bf16 [N, 1024], per-edge weights and / or a ReLU gate, no neighbour codes: the backward pass's
    transposed gather) against csr_gather_kernel: rows that ONE wave sums in edge order there (<= 12 edges) and the listed rows
    (> 24 edges, one workgroup each, in the launch) are equal bit for bit; rows of 13 .. 24 edges are summed by four cooperating
    waves there and in edge order here -- equal to rounding, and both within bf16 output rounding of fp64."""
    from egopack_amd import data as D
    g = gen(77 + 2 * weighted + gated)
    parts, n = [], 0
    for rep in range(60):
        T = 32
        if rep % 4 == 0:  # an LTA sequence: forecast nodes fed by the last input nodes (fan-out rows in the transposed graph)
            y = torch.stack([torch.randint(1, 5, (T,), generator=g), torch.randint(0, 5, (T,), generator=g)], 1)
            y[:T - (20 if rep % 8 else 30)] = -1  # 20 forecast nodes behind a radius of 1.5, or 30 behind the whole sequence
            parts.append(D.lta_connectivity_edges(torch.arange(T), y, 1.5 if rep % 8 else float(T)) + n)
        else:
            parts.append(D.radius_band_edges(torch.arange(T), 1) + n)
        n += T
    if split:  # listed rows beyond the in-launch limit (100 forecast nodes behind every input node): produced by the split launches
        T2 = 256
        y2 = torch.stack([torch.randint(1, 5, (T2,), generator=g), torch.randint(0, 5, (T2,), generator=g)], 1)
        y2[:T2 - 100] = -1
        parts.append(D.lta_connectivity_edges(torch.arange(T2), y2, float(T2)) + n)
        n += T2
    n += 3  # isolated rows
    graph = D.build_csr(torch.cat(parts, 1), n).to(DEV)
    deg = (graph.t_rowptr[1:] - graph.t_rowptr[:-1]).cpu()
    assert (deg > 24).any() and ((deg > 12) & (deg <= 24)).any() and (deg == 0).any() and graph.t_heavy is not None
    assert graph.t_heavy_mode == (0 if split else 1)
    x = torch.randn(n, 1024, generator=g).to(DEV).to(BF)
    gate = torch.randn(n, 1024, generator=g).to(DEV).to(BF) if gated else None
    wgt = graph.t_wgt if weighted else None
    outs = []
    for v2 in (True, False):
        with _rows_v2(v2):
            o = torch.full_like(x, 7.0)
            ops._csr_gather(x, graph.t_rowptr, graph.t_col, wgt, gate, o, graph.t_heavy, graph.t_heavy_mode)
            outs.append(o)
    one_wave = ((deg <= 12) | (deg > 24)).to(DEV)
    assert torch.equal(outs[0][one_wave], outs[1][one_wave])
    torch.testing.assert_close(outs[0].float(), outs[1].float(), rtol=1e-2, atol=1e-2)
    xs, rp, cl = x.double().cpu(), graph.t_rowptr.cpu(), graph.t_col.cpu().long()
    w = graph.t_wgt.double().cpu() if weighted else None
    ref = torch.zeros(n, 1024, dtype=torch.float64)
    for r in range(n):
        e0, e1 = int(rp[r]), int(rp[r + 1])
        if e1 > e0:
            rows_ = xs[cl[e0:e1]]
            ref[r] = (rows_ * w[e0:e1, None]).sum(0) if weighted else rows_.mean(0)
    if gated:
        ref = torch.where(gate.double().cpu() > 0, ref, torch.zeros_like(ref))
    torch.testing.assert_close(outs[0].float().cpu(), ref.float(), **OUT16)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("lens,cols", [((256,) * 16, 1024), ((32,) * 64, 1024), ((1, 0, 7, 300, 33), 260), ((5, 9), 250),
                                       ((32, 0, 9, 40) * 40, 512)])
def test_segment_max_rows_shared_by_four_lanes(ops, lens, cols, dt):
    """The per-sequence max pool with a segment's rows shared by four row lanes -- sixteen when the launch has fewer than 128
    segments -- (egk_segment_max_fwd; BASELINE config 5 pools 16 sequences of 256 nodes): values AND arg-max rows equal the serial
    first-occurrence rule -- ties (bf16 values repeat a lot at
    T = 256) go to the smallest row -- empty segments give zeros / -1, widths that are not a multiple of 4 take the
    thread-per-column kernel."""
    g = gen(sum(lens) + cols)
    n = sum(lens)
    x = (torch.randn(n, cols, generator=g) * 0.5).to(dt)
    x[torch.rand(n, cols, generator=g) < 0.3] = 0.25  # plenty of exact ties
    ptr = torch.tensor([0, *torch.tensor(lens).cumsum(0).tolist()], dtype=torch.int32)
    from egopack_amd import _lib
    xd = x.to(DEV)
    out = torch.empty(len(lens), cols, dtype=dt, device=DEV)
    arg = torch.empty(len(lens), cols, dtype=torch.int32, device=DEV)
    rc = _lib.load().egk_segment_max_fwd(ops._stream(), ops._p(xd), ops._p(ptr.to(DEV)), ops._p(out), ops._p(arg), len(lens), cols,
                                         ops._dt(xd))
    assert rc == 0, _lib.last_error()
    ref_v = torch.zeros(len(lens), cols)
    ref_a = torch.full((len(lens), cols), -1, dtype=torch.int32)
    xf = x.float()
    for s_, (a, b) in enumerate(zip(ptr[:-1].tolist(), ptr[1:].tolist())):
        if b > a:
            v, i = xf[a:b].max(dim=0)
            first = (xf[a:b] == v).int().argmax(dim=0)  # first occurrence of the maximum
            ref_v[s_], ref_a[s_] = v, (first + a).int()
    assert torch.equal(out.float().cpu(), ref_v)
    assert torch.equal(arg.cpu(), ref_a)


# ---------------------------------------------------------------------------------------------------------
# nearest prototypes from ONE bf16 product + proven window + exact re-rank (egk_topk_window)
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K,H,k", [(300, 4096, 1024, 4), (70, 257, 128, 8), (64, 64, 64, 4), (33, 1000, 1024, 16), (40, 512, 2048, 4)])
def test_window_search_gives_the_lists_of_the_exact_distances(ops, N, K, H, k):
    """GraphONE.__compute_edges (reference models/graphONE/graphONE.py:119-141) in the bf16 compute modes: the neighbour lists from
    the one-product search (bf16 MFMA product of the rounded operands, per-row error window, exact re-rank of the candidates) must
    be the lists of the exact distances -- including clusters of prototypes that the bf16 product cannot tell apart, exact
    duplicates (ties -> lower index) and features that are exactly representable in bf16 -- and equal the three-product search
    wherever the ranking gap exceeds its own noise."""
    g = gen(N * 7 + K)
    f, bank = torch.randn(N, H, generator=g), torch.randn(K, H, generator=g)
    if K >= 256:
        # rows 0-9 sit next to a cluster of 6 near-identical prototypes (relative spread 2e-3: cosines ~1e-5 apart, two orders
        # below the resolution of the bf16 product) -- the exact ranking inside the cluster decides the list
        for r in range(10):
            j0 = 10 + 7 * r
            for i in range(1, 6):
                bank[j0 + i] = bank[j0] + 2e-3 * torch.randn(H, generator=g)
            f[r] = bank[j0] + 0.5 * torch.randn(H, generator=g)
        bank[200] = bank[199]  # an exact duplicate: the lower index first
        f[10] = bank[199] + 0.3 * torch.randn(H, generator=g)
        f[11] = f[11].to(torch.bfloat16).float()  # rf = 0: the window is the bank's alone
    fd, bd = f.to(DEV), bank.to(DEV)
    cand = torch.zeros(N, dtype=torch.int32, device=DEV)
    with ops.compute_mode("bf16"):
        ops._window_stats["cand"] = cand
        try:
            nn = ops.cosine_topk(fd, bd, k).cpu()
        finally:
            ops._window_stats["cand"] = None
        prev = ops._window_search["on"]
        ops._window_search["on"] = False
        try:
            nn3 = ops.cosine_topk(fd, bd, k).cpu()
        finally:
            ops._window_search["on"] = prev
    fn, bn = f.double() / f.double().norm(dim=1, keepdim=True), bank.double() / bank.double().norm(dim=1, keepdim=True)
    dist = 1.0 - fn @ bn.T
    srt, order = torch.sort(dist, dim=1, stable=True)
    gap = (srt[:, 1:k + 1] - srt[:, :k]).min(dim=1).values
    cand = cand.cpu()
    assert int(cand.min()) >= k and float(cand.float().mean()) < max(64, 4 * k), (int(cand.min()), float(cand.float().mean()))
    safe = gap > 2e-6  # (the f32 key 1 - dot * f_inv * b_inv carries ~1e-7 of rounding: below that a tie is a tie)
    if K >= 256:
        assert int(safe[:10].sum()) >= 1  # planted clusters that ARE resolvable in f32 -- and must be resolved (checked below)
        assert nn[10, :2].tolist() == [199, 200]
    assert torch.equal(nn[safe], order[safe][:, :k])
    torch.testing.assert_close(torch.gather(dist, 1, nn), srt[:, :k], rtol=0, atol=1e-6)
    safe3 = gap > 1e-5  # the three-product search: exact where the gap exceeds ITS noise
    assert torch.equal(nn[safe3], nn3[safe3])


@pytest.mark.parametrize("G,N,K,H,k", [(3, 128, 4096, 1024, 4), (2, 64, 257, 128, 8), (4, 192, 512, 256, 4)])
def test_grouped_window_search_equals_the_searches_one_by_one(ops, G, N, K, H, k):
    """``ops.nearest_prototypes_grouped`` (the auxiliary tasks of one EgoPack batch as one chain of grouped launches): the lists of
    ``nearest_prototypes`` per task, bit for bit, and the bf16 rounding of the rows as a by-product."""
    g = gen(G * 1000 + N + K)
    base = torch.randn(G * N, H, generator=g).to(DEV)
    banks = [torch.randn(K, H, generator=g).to(DEV) for _ in range(G)]
    for b in banks:
        b[7] = b[6]  # exact duplicates: ties -> lower index
    base[3] = banks[0][6] + 0.2 * base[3]
    feats = [base[i * N:(i + 1) * N] for i in range(G)]
    with ops.compute_mode("bf16"):
        assert ops.nearest_prototypes_grouped_ok(feats, banks, k)
        norms = [ops.row_inv_norm(b) for b in banks]
        lists, hi = ops.nearest_prototypes_grouped(feats, banks, k, norms)
        one = [ops.nearest_prototypes(f, b, k, "cosine", n) for f, b, n in zip(feats, banks, norms)]
        assert not ops.nearest_prototypes_grouped_ok([base[:N], base[2 * N:3 * N]] if G > 2 else [base[:N].clone(), base[N:]], banks[:2], k)
    for a, b in zip(lists, one):
        assert torch.equal(a, b)
    assert lists[0][3, :2].tolist() == [6, 7]
    assert torch.equal(hi, base.to(torch.bfloat16))
    with ops.compute_mode("f32"):
        assert not ops.nearest_prototypes_grouped_ok(feats, banks, k)  # the exact-f32 search stays one per task


def test_window_search_with_bf16_features_and_ties(ops):
    """bf16 features (hi(f) = f: no row residual) and a bank with whole groups of identical rows: ties go to the lower index."""
    bank = torch.randn(64, 64, generator=gen(5))
    bank[1::2] = bank[0::2]  # 32 pairs of identical prototypes
    f = (bank[[4, 10, 20]] + 0.1 * torch.randn(3, 64, generator=gen(6))).to(torch.bfloat16)
    with ops.compute_mode("bf16"):
        nn = ops.cosine_topk(f.to(DEV), bank.to(DEV), 2).cpu()
    assert nn.tolist() == [[4, 5], [10, 11], [20, 21]]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n_src,n_seg,T,cols", [(4, 64, 32, 1024), (2, 200, 5, 256), (3, 7, 33, 64)])
def test_segment_max_of_several_inputs_in_one_launch(ops, dtype, n_src, n_seg, T, cols):
    """``ops.segment_max_multi`` (the OSCC head's pools of its own features and of one GraphONE output per auxiliary task,
    reference models/tasks/oscc.py:68,85): values, winners' gradients and zeros elsewhere equal the pools one by one, bit for bit."""
    g = gen(n_src * 100 + n_seg)
    ptr = torch.arange(0, (n_seg + 1) * T, T, dtype=torch.int32, device=DEV)
    xs = [torch.randn(n_seg * T, cols, generator=g).to(DEV).to(dtype) for _ in range(n_src)]
    xs[0][1] = xs[0][0]  # a tie inside the first sequence: the first occurrence wins
    dys = [torch.randn(n_seg, cols, generator=g).to(DEV).to(dtype) for _ in range(n_src)]
    a = [x.clone().requires_grad_(True) for x in xs]
    b = [x.clone().requires_grad_(True) for x in xs]
    with ops.compute_mode("bf16" if dtype == torch.bfloat16 else "f32"):
        ya = ops.segment_max_multi(a, ptr)
        yb = [ops.segment_max(x, ptr) for x in b]
        torch.autograd.backward(ya, dys)
        torch.autograd.backward(yb, dys)
    for i in range(n_src):
        assert torch.equal(ya[i], yb[i])
        assert torch.equal(a[i].grad, b[i].grad)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1031, 4096), (64, 2048), (5, 3072)])
def test_wide_row_layernorm_kernels_against_the_one_wave_kernels(ops, dt, rows, cols):
    """Rows of 2048 / 3072 / 4096 columns (the shipped temporal pooling is 4096 wide, reference configs/model/temporal_pooling/
    trn.yaml:3) run on workgroup-per-row kernels (egk_tune 7): the SAME dropout keep masks as the one-wave-per-row kernels (the
    Philox counter of a column group does not depend on who draws it), outputs / input gradients equal up to the last bits of the
    row statistics' summation order, parameter gradients up to the order the rows are summed in."""
    from egopack_amd import _lib
    lib = _lib.load()
    g = gen(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2 + 0.3).to(DEV).to(dt)
    w, b = torch.randn(cols, generator=g).to(DEV), torch.randn(cols, generator=g).to(DEV)
    wt = torch.randn(rows, cols, generator=g).to(DEV).to(dt)

    def run(wide):
        prev = lib.egk_tune(7, int(wide))
        try:
            ops.manual_seed(77)
            xi, wi, bi = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            with ops.compute_mode("bf16" if dt == torch.bfloat16 else "f32"):
                y = ops.row_layernorm(xi, wi, bi, 1e-5, relu=True, p=0.25, training=True)
                mask = ops.last_rowln_mask(y).clone()
                (y.float() * wt.float()).sum().backward()
            return y.detach().float(), mask, xi.grad.float(), wi.grad.clone(), bi.grad.clone()
        finally:
            lib.egk_tune(7, prev)
    y1, m1, dx1, dw1, db1 = run(True)
    y0, m0, dx0, dw0, db0 = run(False)
    assert torch.equal(m1, m0) and 0.7 < float(m1.float().mean()) < 0.8
    tol = dict(rtol=2e-2, atol=2e-2) if dt == torch.bfloat16 else dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y1, y0, **tol)
    torch.testing.assert_close(dx1, dx0, **(tol if dt == torch.bfloat16 else dict(rtol=1e-4, atol=1e-4)))
    torch.testing.assert_close(dw1, dw0, rtol=1e-3, atol=2e-3 * rows ** 0.5 if dt == torch.float32 else 5e-2 * rows ** 0.5)
    torch.testing.assert_close(db1, db0, rtol=1e-3, atol=2e-3 * rows ** 0.5 if dt == torch.float32 else 5e-2 * rows ** 0.5)


def test_f16_screen_of_the_grouped_search_keeps_the_lists_and_shrinks_the_window(ops):
    """The grouped search's one-product screen on the f16 matrix instructions (ops._window_f16; egk_cast_f16, egk_gemm_grouped
    with op_f16, egk_topk_window_group16): the SAME lists as the bf16 screen -- both windows are proven, the candidates' distances
    are exact either way -- with far fewer candidates on a bank of near-duplicate prototypes; rows with values beyond the half
    range (their f16 image is inf: an unbounded window) are still ranked exactly."""
    G, N, K, H, k = 3, 256, 2048, 1024, 4
    g = gen(4242)
    centres = torch.randn(64, H, generator=g)
    banks = [(centres[torch.randint(0, 64, (K,), generator=g)] + 0.3 * torch.randn(K, H, generator=g)).to(DEV) for _ in range(G)]
    base = (centres[torch.randint(0, 64, (G * N,), generator=g)] + 0.2 * torch.randn(G * N, H, generator=g)).to(DEV)
    base[5] *= 3.0e4  # |x| up to ~1e5: beyond the half range
    feats = [base[i * N:(i + 1) * N] for i in range(G)]
    cand = torch.zeros(G * N, dtype=torch.int32, device=DEV)
    out = {}
    with ops.compute_mode("bf16"):
        norms = [ops.row_inv_norm(b) for b in banks]
        for f16 in (True, False):
            prev = ops._window_f16["on"]
            ops._window_f16["on"], ops._window_stats["cand"] = f16, cand
            try:
                lists, _ = ops.nearest_prototypes_grouped(feats, banks, k, norms)
                torch.cuda.synchronize()
                out[f16] = (torch.cat(lists).cpu(), cand.clone().cpu())
            finally:
                ops._window_f16["on"], ops._window_stats["cand"] = prev, None
    assert torch.equal(out[True][0], out[False][0])
    c16, cbf = out[True][1].float(), out[False][1].float()
    keep = torch.ones(G * N, dtype=torch.bool)
    keep[5] = False
    assert float(c16[keep].mean()) < 0.6 * float(cbf[keep].mean()), (float(c16[keep].mean()), float(cbf[keep].mean()))
    assert int(c16[5]) == K  # the overflowing row: every prototype is a candidate
    fn, bn = base.double().cpu(), banks[0].double().cpu()
    d = 1.0 - (fn[:N] / fn[:N].norm(dim=1, keepdim=True)) @ (bn / bn.norm(dim=1, keepdim=True)).T
    srt, order = torch.sort(d, dim=1, stable=True)
    safe = (srt[:, 1:k + 1] - srt[:, :k]).min(dim=1).values > 2e-6
    assert bool(safe[5]) and torch.equal(out[True][0][:N][safe], order[safe][:, :k])
