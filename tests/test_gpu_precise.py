"""The f32-grade feature path on the bf16 matrix pipe ('bf16x3': x = hi + lo, a.b ~ a_hi.b_hi + a_hi.b_lo + a_lo.b_hi) that feeds
the nearest-prototype index op (reference models/graphONE/graphONE.py:119-141) when the training pass stores bf16 activations.

  * egk_split_bf16: hi / lo bit-exact against torch's round-to-nearest-even casts;
  * contractions with up to six K sources == the sum of the separate contractions (exact-f32 and bf16 kernels);
  * three-product contraction against fp64: relative Frobenius error <= 3e-5 (plain bf16: ~3e-3; stated so that a broken low
    half -- error back at 2^-9 -- cannot pass), also through Linear / SAGE layers with the optimizer's flat buffers;
  * backbone + projection in 'bf16x3' mode against the CPU oracle (f32): features within 2e-4 relative Frobenius;
  * nearest-prototype lists in 'bf16' mode == the exact-f32 lists wherever the f32 ranking gap exceeds 1e-5.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import path as O  # noqa: E402
from oracle import pyg_ops as P  # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd import ops as _ops
    return _ops


def gen(seed):
    return torch.Generator().manual_seed(seed)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))


@pytest.mark.parametrize("rows,cols,ld,want_hi", [(1, 1 << 16, 1 << 16, True), (37, 100, 104, True), (64, 1024, 1024, False), (5, 7, 9, True)])
def test_split_bf16_is_the_two_rounded_halves(ops, rows, cols, ld, want_hi):
    from egopack_amd import _lib
    x = torch.randn(rows, ld, generator=gen(rows + cols)) * 3.0
    x[0, 0] = 0.0
    xd = x.to(DEV)
    hi = torch.zeros(rows, cols, dtype=torch.bfloat16, device=DEV) if want_hi else None
    lo = torch.zeros(rows, cols, dtype=torch.bfloat16, device=DEV)
    rc = _lib.load().egk_split_bf16(ops._stream(), ops._p(xd), ld, ops._p(hi), ops._p(lo), cols, rows, cols)
    assert rc == 0, _lib.last_error()
    ref_hi = x[:, :cols].to(torch.bfloat16)
    ref_lo = (x[:, :cols] - ref_hi.float()).to(torch.bfloat16)
    if want_hi:
        assert torch.equal(hi.cpu(), ref_hi)
    assert torch.equal(lo.cpu(), ref_lo)
    # hi + lo carries 16 significant bits of x
    assert _rel(ref_hi.double() + ref_lo.double(), x[:, :cols]) < 2e-5


@pytest.mark.parametrize("compute,Ks", [("bf16", (64, 128, 64, 192, 64, 64)), ("bf16", (128, 64, 64)), ("f32", (40, 72, 16, 32)),
                                        ("bf16", (40, 72, 24))])
def test_gemm_extra_k_sources_equal_the_sum_of_the_parts(ops, compute, Ks):
    """C = sum_s A_s . B_s^T with 3 .. 6 sources in ONE launch (pipelined kernel when every K is a multiple of 64, generic
    kernel otherwise) against fp64 on the operand values the kernel reads."""
    from egopack_amd import _lib
    import ctypes as C
    g = gen(len(Ks) * 17 + Ks[0])
    M, N = 200, 136
    dt = torch.bfloat16 if compute == "bf16" else torch.float32
    As = [torch.randn(M, k, generator=g).to(dt).to(DEV) for k in Ks]
    Bs = [torch.randn(N, k, generator=g).to(dt).to(DEV) for k in Ks]
    bias = torch.randn(N, generator=g).to(DEV)
    out = torch.empty(M, N, device=DEV)
    d = ops._gemm_desc(M, N, As[0], Ks[0], Bs[0], Ks[0], Ks[0], out, N, A2=As[1], lda2=Ks[1], B2=Bs[1], ldb2=Ks[1], K2=Ks[1],
                       bias=bias, compute=ops.BF16 if compute == "bf16" else ops.F32)
    d.n_extra = len(Ks) - 2
    for i in range(2, len(Ks)):
        d.xA[i - 2], d.xB[i - 2], d.xlda[i - 2], d.xldb[i - 2], d.xK[i - 2] = As[i].data_ptr(), Bs[i].data_ptr(), Ks[i], Ks[i], Ks[i]
    rc = _lib.load().egk_gemm(ops._stream(), C.byref(d))
    assert rc == 0, _lib.last_error()
    ref = sum(a.double().cpu() @ b.double().cpu().t() for a, b in zip(As, Bs)) + bias.double().cpu()
    tol = dict(rtol=1e-3, atol=1e-3) if compute == "bf16" else dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out.cpu().double(), ref, **tol)


@pytest.mark.parametrize("M,N,K,K2", [(2048, 1024, 1024, 0), (300, 256, 4608, 0), (2048, 1024, 1024, 1024), (130, 70, 40, 0)])
def test_three_product_contraction_is_f32_grade(ops, M, N, K, K2):
    """gemm(compute = X3) on f32 operands against fp64: <= 3e-5 relative Frobenius where a plain bf16 contraction of the same
    operands is at ~3e-3 (asserted, so the comparison means something).  The last shape does not qualify for the pipelined
    kernel (K % 64 != 0) and runs on the exact-f32 instructions instead."""
    g = gen(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    A2 = torch.randn(M, K2, generator=g) if K2 else None
    B2 = torch.randn(N, K2, generator=g) if K2 else None
    bias = torch.randn(N, generator=g)
    ref = A.double() @ B.double().t() + bias.double()
    if K2:
        ref = ref + A2.double() @ B2.double().t()
    kw = dict(A2=A2.to(DEV), lda2=K2, B2=B2.to(DEV), ldb2=K2, K2=K2) if K2 else {}
    out = torch.empty(M, N, device=DEV)
    ops.gemm(M, N, A.to(DEV), K, B.to(DEV), K, K, out, N, bias=bias.to(DEV), compute=ops.X3, **kw)
    err = _rel(out.cpu(), ref)
    assert err < 3e-5, err
    if K % 64 == 0:
        out16 = torch.empty(M, N, device=DEV)
        kw16 = dict(A2=A2.to(DEV).bfloat16(), lda2=K2, B2=B2.to(DEV).bfloat16(), ldb2=K2, K2=K2) if K2 else {}
        ops.gemm(M, N, A.to(DEV).bfloat16(), K, B.to(DEV).bfloat16(), K, K, out16, N, bias=bias.to(DEV), compute=ops.BF16, **kw16)
        assert _rel(out16.cpu(), ref) > 1e-3


@pytest.mark.parametrize("tA,tB", [(False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(1024, 512, 2048), (640, 1024, 6144)])
def test_three_product_contraction_with_k_major_operands(ops, M, N, K, tA, tB):
    """The dX (B stored [K, N]) and dW (A stored [K, M], B stored [K, N], accumulating, with the bias gradient) forms of a
    Linear's backward in three-product mode: operands are split in the layout they are stored in; the bias gradient, which
    the kernel can only fuse for a single K source, becomes a column-sum launch of its own.  <= 3e-5 against fp64."""
    g = gen(M + 3 * N + K + tA + 2 * tB)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((K, N) if tB else (N, K), generator=g)
    ref = (A.double().t() if tA else A.double()) @ (B.double() if tB else B.double().t())
    out = torch.randn(M, N, generator=g)
    ref = ref + out.double()
    out = out.to(DEV)
    kw = {}
    if tA and tB:
        db = torch.randn(M, generator=g)
        ref_db = db.double() + A.double().sum(0)
        kw["dbias"] = db = db.to(DEV)
    ops.gemm(M, N, A.to(DEV), A.shape[1], B.to(DEV), B.shape[1], K, out, N, transA=tA, transB=tB, accumulate=True, compute=ops.X3, **kw)
    assert _rel(out.cpu(), ref) < 3e-5, _rel(out.cpu(), ref)
    if kw:
        assert _rel(db.cpu(), ref_db) < 1e-6


def test_linear_backward_in_three_product_mode_is_f32_grade(ops):
    """A Linear's forward + backward under compute mode 'bf16x3' (f32 activations, every contraction three bf16 products):
    output, input gradient, weight and bias gradient against fp64 <= 3e-5 -- the mode as a TRAINING mode, not only the
    forward-only feature pass."""
    from egopack_amd.models.layers import Linear
    torch.manual_seed(5)
    lin = Linear(1024, 512).to(DEV)
    x = torch.randn(2048, 1024, device=DEV, requires_grad=True)
    cot = torch.randn(2048, 512, device=DEV)
    with ops.compute_mode("bf16x3"):
        y = lin(x)
        (y * cot).sum().backward()
        ops.join_wgrad(force=True)
    torch.cuda.synchronize()
    xd, wd, bd, cd = x.detach().double().cpu(), lin.weight.detach().double().cpu(), lin.bias.detach().double().cpu(), cot.double().cpu()
    assert _rel(y.detach().cpu(), xd @ wd.t() + bd) < 3e-5
    assert _rel(x.grad.cpu(), cd @ wd) < 3e-5
    assert _rel(lin.weight.grad.cpu(), cd.t() @ xd) < 3e-5
    assert _rel(lin.bias.grad.cpu(), cd.sum(0)) < 1e-6


def test_three_product_linear_uses_the_optimizers_halves(ops):
    """A Linear whose weight lives in FlatAdam's flat buffers: hi = the bf16 shadow the Adam kernel keeps, lo = the flat low
    buffer -- refreshed on demand the first time, then kept by the Adam launch itself (round 5; with ``adam_writes_lo`` off:
    stale after every optimizer launch and refreshed by the next use)."""
    from egopack_amd.models.layers import Linear
    from egopack_amd.optim import FlatAdam
    torch.manual_seed(3)
    lin = Linear(1024, 512).to(DEV)
    opt = FlatAdam(list(lin.parameters()), lr=1e-2)
    x = torch.randn(256, 1024, device=DEV)
    with ops.compute_mode("f32"):
        lin(x).sum().backward()
    opt.step()  # materialises the flat buffers, writes the shadows
    assert getattr(lin.weight, "_egk_shadow", None) is not None

    def run():
        with torch.no_grad(), ops.precise_scope():
            return lin(x)
    ref = lambda: (x.double().cpu() @ lin.weight.detach().double().cpu().t() + lin.bias.detach().double().cpu())
    assert _rel(run().cpu(), ref()) < 3e-5
    off, _ = opt._slot_of[id(lin.weight)]
    assert opt.flat_w16lo is not None and opt._lo_is_fresh(off, lin.weight.numel())
    lin.weight.grad.normal_()
    opt.step()  # parameters move: the Adam launch wrote the new low halves itself
    assert opt._lo_is_fresh(off, lin.weight.numel())
    assert _rel(run().cpu(), ref()) < 3e-5
    opt.adam_writes_lo = False  # (instance override: the round-4 behaviour)
    lin.weight.grad.normal_()
    opt.step()  # parameters move: the low halves are stale and must be refreshed by the next use
    assert not opt._lo_fresh
    assert _rel(run().cpu(), ref()) < 3e-5


def test_a_bf16_input_of_the_precise_pass_is_widened_only_on_demand(ops, monkeypatch):
    """``ops.to_act(x, lazy=True)`` inside ``precise_scope``: an f32 tensor that is never written while only three-product
    contractions consume it (they read the bf16 source), the same bits as the eagerly widened input, and written the moment
    anything else takes its pointer."""
    from egopack_amd.models.layers import Linear
    torch.manual_seed(21)
    lin = Linear(1536, 512).to(DEV)
    x = torch.randn(192, 1536, device=DEV).to(torch.bfloat16)
    with torch.no_grad(), ops.compute_mode("bf16"), ops.precise_scope():
        lazy = ops.to_act(x, lazy=True)
        assert lazy.dtype == torch.float32 and getattr(lazy, "_egk_virtual", False)
        lazy.fill_(float("nan"))  # whatever the allocation held: the contraction must not read it
        y_lazy = lin(lazy)
        assert lazy._egk_virtual  # still not written
        y_eager = lin(ops.to_act(x))
        assert torch.equal(y_lazy, y_eager) and torch.isfinite(y_lazy).all()
        c = ops._c(lazy)  # any other consumer: written first
        assert not lazy._egk_virtual and torch.equal(c, x.float())
        monkeypatch.setenv("EGK_DISABLE", "x3_lazy_input")
        assert not getattr(ops.to_act(x, lazy=True), "_egk_virtual", False)
    with torch.no_grad(), ops.compute_mode("bf16"):
        assert ops.to_act(x, lazy=True) is x  # outside the precise pass: the mode's own element type


def test_backbone_in_three_product_mode_matches_the_f32_oracle(ops, golden):
    """Graph.forward + a projection head under ``precise_scope`` (bf16 input features, as the bf16 training pass stores them)
    against the CPU oracle in f32: the features that rank the prototypes.  H = 256 so that every contraction takes the
    pipelined three-product path."""
    from egopack_amd import data as D
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import RecognitionTask
    torch.manual_seed(11)
    F_IN, S, H = 64, 3, 256  # S * F_IN = 192: a multiple of 64
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
    model = Graph(F_IN, hidden_size=H, depth=3, temporal_pooling=trn, num_segments=S)
    task = RecognitionTask(H, H, (13, 17))
    sd, tsd = {k: v.clone() for k, v in model.state_dict().items()}, {k: v.clone() for k, v in task.state_dict().items()}
    ds = D.SyntheticTaskDataset("ar", 8, 16, S, F_IN, (13, 17), k=1, seed=5)
    host = D.collate([ds[i] for i in range(8)])
    host.x = host.x.to(torch.bfloat16)  # bf16-representable features
    od = P.OData(x=host.x.float(), pos=host.pos, edge_index=host.edge_index, batch=host.batch, y=host.y, num_graphs=host.num_graphs)
    with torch.no_grad():
        ref = O.projection_features(tsd, O.graph_forward(sd, od.x, od.pos, od.edge_index, S))
    model.to(DEV).eval()
    task.to(DEV).eval()
    dev = host.to(DEV)
    with torch.no_grad(), ops.compute_mode("bf16"):
        with ops.precise_scope():
            got = task.forward_features(model(dev), out_f32=True)
        plain = task.forward_features(model(dev), out_f32=True)
    assert got.dtype == torch.float32
    e3, e1 = _rel(got.cpu(), ref), _rel(plain.float().cpu(), ref)
    assert e3 < 2e-4, (e3, e1)
    assert e1 > 10 * e3, (e3, e1)  # (the plain bf16 pass is an order of magnitude further away: the pass is worth its cost)


def test_nearest_prototypes_in_bf16_mode_are_the_exact_lists(ops):
    """The search product as three bf16 products (every mode but 'f32') against the exact-f32 product: identical lists wherever
    the fp64 ranking gap exceeds 1e-5, K = 4096 / H = 1024 / k = 4, cosine and l2."""
    g = torch.Generator(device=DEV).manual_seed(2)
    f = torch.randn(2048, 1024, device=DEV, generator=g)
    bank = torch.nn.Parameter(torch.randn(4096, 1024, device=DEV, generator=g), requires_grad=False)
    for dist in ("cosine", "l2"):
        with ops.compute_mode("f32"):
            exact = ops.nearest_prototypes(f, bank, 4, dist)
        with ops.compute_mode("bf16"):
            fast = ops.nearest_prototypes(f, bank, 4, dist)
            again = ops.nearest_prototypes(f, bank, 4, dist)  # (second call: the bank's halves come from the cache)
        assert torch.equal(fast, again)
        fd, bd = f.double(), bank.detach().double()
        if dist == "cosine":
            d = 1 - (fd / fd.norm(dim=1, keepdim=True)) @ (bd / bd.norm(dim=1, keepdim=True)).t()
        else:
            d = torch.cdist(fd, bd) / 4096
        srt = d.sort(dim=-1).values
        # (l2 keys are cdist / 4096 ~ 0.011: a gap of 1e-7 there is a gap of ~0.02 in the dot product, ~100x the product's error)
        safe = (srt[:, 1:5] - srt[:, :4]).min(dim=1).values > (1e-5 if dist == "cosine" else 1e-7)
        assert safe.float().mean() > 0.97
        assert torch.equal(fast[safe], exact[safe]), dist
        assert float((fast == exact).float().mean()) > 0.999


@pytest.mark.parametrize("cols", [1024, 250])
def test_split_tee_stores_the_halves_of_the_split_launch(ops, cols, monkeypatch):
    """egk_tee_split_next: inside a precise scope the row kernels that PRODUCE a contraction operand (row LayerNorm, graph
    LayerNorm, positional-encoding add, mean gather) also store its bf16 halves -- bit for bit what egk_split_bf16 makes of
    their f32 result -- so that the contraction needs no split launch; with EGK_DISABLE=x3_tee the same calls leave no halves
    behind and the results are the same bits."""
    from egopack_amd import data as D
    g = gen(cols)
    rows = 96
    x = (torch.randn(rows, cols, generator=g) * 2).to(DEV)
    w, b = torch.randn(cols, generator=g).to(DEV), torch.randn(cols, generator=g).to(DEV)
    seg = torch.tensor([0, 40, rows], dtype=torch.int32, device=DEV)
    ei = torch.cat([D.radius_band_edges(torch.arange(32), 1) + 32 * i for i in range(3)], 1)
    graph = D.build_csr(ei, rows).to(DEV)
    pos = (torch.arange(rows) % 32 - 16).to(DEV)
    freq = P.positional_encoding_frequency(cols).to(DEV)
    calls = {
        "rowln": lambda: ops.row_layernorm(x, w, b, 1e-5, relu=True),
        "graphln": lambda: ops.graph_layernorm_lrelu(x, w, b, seg, 1e-5, 0.2),
        "pe": lambda: ops.pe_add(x, pos, freq, (-16, 15)),
        "pe_direct": lambda: ops.pe_add(x, pos, freq),
        "gather": lambda: ops.csr_mean_aggregate(x, graph),
    }
    for name, fn in calls.items():
        outs = {}
        for tee in (True, False):
            if tee:
                monkeypatch.delenv("EGK_DISABLE", raising=False)
            else:
                monkeypatch.setenv("EGK_DISABLE", "x3_tee")
            with torch.no_grad(), ops.precise_scope():
                y = fn()
                key = (y.data_ptr(), rows, cols, cols, y._version)
                hit = ops._x3["cache"].get(key)
                assert (hit is not None) is tee, (name, tee)
                hi, lo, ld = ops._x3_act(y, rows, cols, cols)  # (tee off: this is where the split launch happens)
                outs[tee] = (y.clone(), hi.clone(), lo.clone())
        torch.cuda.synchronize()
        for a, b_ in zip(outs[True], outs[False]):
            assert torch.equal(a, b_), name
        y, hi, lo = outs[True]
        assert torch.equal(hi, y.to(torch.bfloat16)) and torch.equal(lo, (y - hi.float()).to(torch.bfloat16)), name
    monkeypatch.delenv("EGK_DISABLE", raising=False)
    # outside a scope nothing is armed, and an armed tee refuses a bf16 result
    y = ops.row_layernorm(x, w, b, 1e-5, relu=True)
    assert ops._x3["cache"] is None
    from egopack_amd import _lib
    hi, lo = (torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    assert _lib.load().egk_tee_split_next(ops._p(hi), ops._p(lo), cols) == 0
    with pytest.raises(RuntimeError, match="split tee"):
        ops.row_layernorm(x.to(torch.bfloat16), w, b, 1e-5, relu=True)
    assert _lib.load().egk_tee_split_next(None, None, 0) == 0


def test_adam_launch_keeps_the_low_halves_of_the_weight_operands():
    """Once the low halves of the three-product contractions' weight operands exist (FlatAdam.ensure_lo_shadows), the Adam launch
    writes them for the slice it updates (egk_adam_step_bump): after a step they equal egk_split_bf16 of the updated parameters
    bit for bit, are marked fresh (no split launch at the head of the next precise pass), and a partial update leaves the other
    slices' state as it was."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd import _lib, ops
    from egopack_amd.optim import FlatAdam
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64, 96, device="cuda")), torch.nn.Parameter(torch.randn(40, device="cuda")),
          torch.nn.Parameter(torch.randn(128, 64, device="cuda"))]
    opt = FlatAdam(ps, lr=1e-2, weight_decay=1e-4)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()  # (materialises the flat buffers)
    assert opt.flat_w16lo is None
    assert opt.ensure_lo_shadows() and not opt._lo_is_fresh(0, opt.flat_p.numel())
    opt.flat_g.normal_()
    opt.step()
    n = opt.flat_p.numel()
    assert opt._lo_is_fresh(0, n)
    want_lo = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    assert _lib.load().egk_split_bf16(ops._stream(), ops._p(opt.flat_p), n, None, ops._p(want_lo), n, 1, n) == 0
    assert torch.equal(opt.flat_w16lo, want_lo)
    assert torch.equal(opt.flat_w16, opt.flat_p.to(torch.bfloat16))
    # a partial update: its slice is fresh and right, a foreign write to the parameters invalidates everything
    opt.flat_g.normal_()
    opt.prepare_hyper()
    opt.launch(None, 0, 64 * 96)
    assert opt._lo_is_fresh(0, 64 * 96) and opt._lo_is_fresh(0, n)
    assert _lib.load().egk_split_bf16(ops._stream(), ops._p(opt.flat_p), n, None, ops._p(want_lo), n, 1, n) == 0
    assert torch.equal(opt.flat_w16lo, want_lo)
    # an out-of-band write to the parameters (checkpoint load, a restored snapshot): refresh_shadows rebuilds BOTH copies -- a captured
    # step whose Adam launches keep the low halves holds no split launch that would
    with torch.no_grad():
        opt.flat_p.mul_(1.5)
    opt.refresh_shadows()
    assert opt._lo_is_fresh(0, n)
    assert _lib.load().egk_split_bf16(ops._stream(), ops._p(opt.flat_p), n, None, ops._p(want_lo), n, 1, n) == 0
    assert torch.equal(opt.flat_w16lo, want_lo) and torch.equal(opt.flat_w16, opt.flat_p.to(torch.bfloat16))
    opt.refresh_shadows(8, 64)  # (a range: the gathered slices of a sharded update)
    assert opt._lo_is_fresh(0, n) and torch.equal(opt.flat_w16lo, want_lo)


def test_one_pass_backbone_graph_from_the_precise_passs_tape(ops):
    """ops.dual_record / dual_replay on Graph.forward (TRN pooling, three SAGE layers): the replayed bf16 graph's output IS the
    rounding of the precise pass's output, every taped node is consumed, the backward of the replayed graph gives the gradients of
    the plain bf16 pass up to bf16 noise, and a replay against another node sequence is refused."""
    from egopack_amd import data as D
    from egopack_amd.models import Graph
    torch.manual_seed(12)
    F_IN, S, H = 64, 3, 256
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
    model = Graph(F_IN, hidden_size=H, depth=3, temporal_pooling=trn, num_segments=S).to(DEV).eval()
    ds = D.SyntheticTaskDataset("ar", 8, 16, S, F_IN, (13, 17), k=1, seed=5)
    host = D.collate([ds[i] for i in range(8)])
    host.x = host.x.to(torch.bfloat16)
    dev = host.to(DEV)
    w = torch.randn(dev.pos.shape[0], H, device=DEV)
    with ops.compute_mode("bf16"):
        with torch.no_grad(), ops.precise_scope(), ops.dual_record() as tape:
            f32 = model(dev)
        kinds = [k for k, _ in tape]
        assert kinds == ["linear", "rowln", "linear", "rowln", "linear", "pe_add"] + ["sage_mean", "graphln"] * 3 + ["linear"], kinds
        with ops.dual_replay(tape):
            f16 = model(dev)
        assert f16.dtype == torch.bfloat16 and f16.requires_grad and torch.equal(f16, f32.to(torch.bfloat16))
        (f16.float() * w).sum().backward()
        got = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
        model.zero_grad()
        plain = model(dev)
        (plain.float() * w).sum().backward()
        want = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
        with pytest.raises(RuntimeError, match="dual_replay"):
            with ops.dual_replay(tape[1:]):
                model(dev)
    model.zero_grad()
    dev.x = dev.x.float()
    with ops.compute_mode("f32"):
        (model(dev) * w).sum().backward()
    ref = {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}
    assert _rel(plain.detach().float().cpu(), f32.cpu()) > _rel(f16.detach().float().cpu(), f32.cpu())
    for k in ref:  # no further from the exact-f32 gradients than the plain bf16 pass is (both carry bf16 rounding in backward)
        if float(ref[k].norm()) > 1e-6:
            e1, e0 = _rel(got[k].cpu(), ref[k].cpu()), _rel(want[k].cpu(), ref[k].cpu())
            assert e1 < max(1.25 * e0, 2e-2), (k, e1, e0)


def test_precise_pass_without_reduce_launches_is_the_same_pass(ops):
    """The precise pass's contractions that split K leave their two slabs to the row kernel that reads the result next (row
    LayerNorm, graph LayerNorm, PE add: egk_gemm_defer_reduce_next / egk_slab_input_next) instead of launching the reduce: same
    arithmetic in the same order ((slab 0 + slab 1) + bias), so the pass's output, every taped tensor and the statistics are the
    SAME BITS as with the reduce launches -- at the size where the policy splits (2048 rows, H = 1024)."""
    from egopack_amd import data as D
    from egopack_amd.models import Graph
    torch.manual_seed(3)
    F_IN, S, H = 256, 3, 1024
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
    model = Graph(F_IN, hidden_size=H, depth=3, temporal_pooling=trn, num_segments=S).to(DEV).eval()
    ds = D.SyntheticTaskDataset("ar", 64, 32, S, F_IN, (13, 17), k=1, seed=5)
    host = D.collate([ds[i] for i in range(64)])
    host.x = host.x.to(torch.bfloat16)
    dev = host.to(DEV)
    taken = []
    real = ops._slab_consumer

    def counting(x, cols_max=1024):
        out = real(x, cols_max)
        taken.append(out[1])
        return out

    def run(on):
        prev = ops._slab_defer["on"]
        ops._slab_defer["on"] = on
        try:
            with ops.compute_mode("bf16"), torch.no_grad(), ops.precise_scope(), ops.dual_record() as tape:
                f = model(dev)
            torch.cuda.synchronize()
            return f.clone(), [(k, {n: t.clone() for n, t in ts.items() if torch.is_tensor(t)}) for k, ts in tape]
        finally:
            ops._slab_defer["on"] = prev
    ops._slab_consumer = counting
    try:
        f1, t1 = run(True)
    finally:
        ops._slab_consumer = real
    f0, t0 = run(False)
    assert sum(taken) >= 4, taken  # (the row LayerNorms behind launches that split K in two, three graph LayerNorms at this size)
    assert torch.equal(f1, f0)
    assert [k for k, _ in t1] == [k for k, _ in t0]
    for (k, a), (_, b) in zip(t1, t0):
        assert a.keys() == b.keys(), k
        for n in a:
            assert torch.equal(a[n], b[n]), (k, n)


def test_slab_aware_row_kernels_equal_the_reduce_launch(ops):
    """ops.linear(slab_ok=True) inside a precise_scope at a size where the launch splits K in two, followed by each slab-aware
    consumer (row LayerNorm, graph LayerNorm + LeakyReLU, PE add): the consumer's result AND the linear's result (which the
    consumer stores on its way) equal the path with the reduce launch, bit for bit; a result that nobody slab-aware reads is
    reduced when anything asks for its pointer, and at the latest when the scope ends (never readable unreduced)."""
    torch.manual_seed(9)
    M, K, N = 2048, 1024, 1024
    x = torch.randn(M, K, device=DEV)
    W, b = torch.randn(N, K, device=DEV) * 0.03, torch.randn(N, device=DEV)
    lw, lb = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    seg = torch.tensor([0, 700, M], dtype=torch.int32, device=DEV)
    pos = torch.arange(M, device=DEV, dtype=torch.int64) % 32
    freq = torch.logspace(0, 1, N // 2, 1e-4).to(DEV)

    def run(kind, slab_ok):
        with ops.compute_mode("bf16"), torch.no_grad(), ops.precise_scope():
            y = ops.linear(x, W, b, slab_ok=slab_ok)
            owed = getattr(y, "_egk_slabs", None) is not None
            if kind == "rowln":
                z = ops.row_layernorm(y, lw, lb, 1e-5, relu=True)
            elif kind == "graphln":
                z = ops.graph_layernorm_lrelu(y, lw, lb, seg, 1e-5, 0.2)
            elif kind == "pe":
                z = ops.pe_add(y, pos, freq, (0, 31))
            elif kind == "pointer":
                z = ops.cast_raw(y, torch.bfloat16)  # (not slab-aware: asks for the pointer -> the reduce launch runs first)
            else:
                z = None  # nobody reads it inside the scope
            still = getattr(y, "_egk_slabs", None) is not None
        assert getattr(y, "_egk_slabs", None) is None  # settled by the scope at the latest
        torch.cuda.synchronize()
        return y.clone(), None if z is None else z.clone(), owed, still
    for kind in ("rowln", "graphln", "pe", "pointer", "escape"):
        y1, z1, owed1, still1 = run(kind, True)
        y0, z0, owed0, still0 = run(kind, False)
        assert owed1 and not owed0, kind  # (the flagged launch split K in two and left its slabs)
        assert still1 == (kind == "escape"), kind
        assert torch.equal(y1, y0), kind
        if z1 is not None:
            assert torch.equal(z1, z0), kind
