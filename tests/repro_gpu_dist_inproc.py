"""ROUND-3 LAYOUT, kept to reproduce the round-3 driver abort (NOT collected by default: run it explicitly, after
tests/test_gpu_blockwise.py and tests/test_gpu_configs.py in ONE pytest process -- tools/round4/repro_abort.sh).  The process group lives
in the pytest process here; the product no longer swallows a failed one-graph capture, so the exception that used to precede
the abort is now reported.

The data-parallel step path on ONE GPU: a 1-rank RCCL process group driven through dist.GradSync as if the world
had 2 ranks (the all-reduce then sums a single contribution), with bf16 gradient compression, eager and hipGraph
replay.  Checks the plumbing the 8-GPU run uses: cast -> all-reduce on the side stream -> Adam reading bf16 grads."""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")  # (the round-3 fixed port, part of the reproduction)
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


def _setup(golden, sync):
    import egopack_amd.data as D
    from egopack_amd import engine
    from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask
    from egopack_amd.optim import FlatAdam
    G = golden("mtl_train")
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": 40}
    model = Graph(48, hidden_size=32, depth=3, temporal_pooling=trn, num_segments=3)
    model.load_state_dict(G["before"]["temporal_graph"])
    tasks = {"ar": RecognitionTask(32, 32, (7, 11)), "oscc": OSCCTask(32, 32), "lta": LTATask(32, 32, (7, 11)), "pnr": PNRTask(32, 32)}
    for t, n in (("ar", "task/recognition"), ("oscc", "task/oscc"), ("lta", "task/lta"), ("pnr", "task/pnr")):
        tasks[t].load_state_dict(G["before"][n])
        tasks[t].cuda()
    model.cuda()

    class DS:
        has_joint_label, num_labels = False, 2
    crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), DS()), "lta": MetricSelectorWrapper(CrossEntropyNone(), DS()),
            "oscc": CrossEntropyNone(), "pnr": BCEWithLogitsNone()}
    live = [*model.parameters(), *(p for t in ("ar", "lta", "pnr") for p in tasks[t].parameters())]
    opt = FlatAdam(live, lr=1e-3, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, G["weights"], opt, fused_backbone=True, sync=sync)
    batches = {}
    for t in ("ar", "lta", "pnr"):
        b = D.Data(**G["batches"][t][0])
        b.graph = D.build_csr(b.edge_index, b.x.shape[0])
        b.ptr32 = b.ptr.to(torch.int32)
        batches[t] = b.to("cuda")
    return step, opt, batches


def test_dp_step_with_bf16_compressed_allreduce(pg, golden):
    from egopack_amd import ops
    from egopack_amd.dist import GradSync
    with ops.compute_mode("f32"):
        # reference: no exchange, gradient scale 1/2 applied by Adam on the f32 gradients
        step, opt, batches = _setup(golden, None)
        opt.grad_scale = 0.5
        for _ in range(3):
            step.step(batches)
        ref = opt.flat_p.clone()
        # DP path, eager: "world of 2" whose all-reduce sums one contribution -> same gradients / 2, bf16-rounded
        step, opt, batches = _setup(golden, GradSync(2, chunk_mb=0.01, compress="bf16"))
        for _ in range(3):
            step.step(batches)
        torch.cuda.synchronize()
        assert opt.grad_scale == 0.5 and step.sync._g16 is not None and step.sync._g16.dtype == torch.bfloat16
        # Adam normalises the update; bf16 rounding of a gradient moves an update by << lr
        torch.testing.assert_close(opt.flat_p, ref, rtol=0, atol=2e-4)
        eager = opt.flat_p.clone()
        # DP path, hipGraph: the whole step incl. the RCCL collectives and the per-chunk Adam launches in ONE graph (default on
        # an RCCL group), or the backward stages captured with exchange + Adam issued between / after the graphs
        for one_graph in (True, False):
            step, opt, batches = _setup(golden, GradSync(2, chunk_mb=0.01, compress="bf16"))
            step.one_graph_exchange = one_graph
            step.capture(batches, warmup=1)
            assert step._graph_has_exchange is one_graph and bool(step._fuse_adam) is one_graph
            assert isinstance(step._graph, list) is (not one_graph) and not getattr(step, "capture_notes", [])
            for _ in range(2):
                step.replay()
            torch.cuda.synchronize()
            assert opt.step_count == 3
            torch.testing.assert_close(opt.flat_p, eager, rtol=0, atol=2e-6)


def test_adam_reads_bf16_gradients(pg):
    from egopack_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(5)
    p0, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    cp = p0.clone().requires_grad_(True)
    cp.grad = gr.to(torch.bfloat16).float()
    ref = torch.optim.Adam([cp], lr=1e-2, weight_decay=1e-3)
    ref.step()
    dp = p0.clone().cuda().requires_grad_(True)
    dp.grad = gr.clone().cuda()
    opt = FlatAdam([dp], lr=1e-2, weight_decay=1e-3)
    opt._materialise()
    opt.step(grads=opt.flat_g.to(torch.bfloat16))
    torch.testing.assert_close(dp.detach().cpu(), cp.detach(), rtol=1e-5, atol=1e-6)
    # the bf16 shadow the contractions read tracks the updated parameters
    torch.testing.assert_close(opt.flat_w16[:1000].float().cpu(), cp.detach().to(torch.bfloat16).float(), rtol=0, atol=0)


@pytest.mark.parametrize("with_sync", [False, True])
def test_staged_backward_equals_the_one_piece_backward(pg, golden, with_sync):
    """The three-stage backward (heads | SAGE stack | TRN, cut at detached leaves, region-wise exchange between the
    stages) produces the parameters of the one-piece backward BIT FOR BIT -- eager and as three captured graphs."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync

    def run(staged, graph, one_graph=False):
        sync = GradSync(2, chunk_mb=0.01, compress="bf16") if with_sync else None
        step, opt, batches = _setup(golden, sync)
        step.staged = staged
        step.one_graph_exchange = one_graph
        if not with_sync:
            opt.grad_scale = 0.5
        if graph:
            step.capture(batches, warmup=1)
            assert isinstance(step._graph, list) == (bool(staged) and not (one_graph and with_sync))
            assert step._graph_has_exchange is bool(staged and one_graph and with_sync)
            for _ in range(2):
                step.replay()
        else:
            for _ in range(3):
                step.step(batches)
        torch.cuda.synchronize()
        assert opt.step_count == 3
        if staged:
            heads, mid, trn = step._stage_regions()
            assert trn[0] == 0 and trn[1] == mid[0] and mid[1] == heads[0] and heads[1] == opt.flat_p.numel()
        return opt.flat_p.clone()
    with ops.compute_mode("f32"):
        ref = run(False, False)
        assert torch.equal(run(True, False), ref)
        assert torch.equal(run(False, True), ref)
        assert torch.equal(run(True, True), ref)
        assert torch.equal(run(True, True, one_graph=True), ref)  # (stages + collectives + Adam slices in ONE captured graph)


def test_exact_graph_ln_mode_is_captured_with_the_exchange(pg, golden):
    """exact_graph_ln (graph-LayerNorm statistics summed over the ranks, forward and backward): on an RCCL group its six small
    collectives per step are captured with the rest of the N-rank step -- the replayed graph gives the parameters of the
    eagerly issued steps bit for bit (1-rank group driven as a world of 2: the sums hold one contribution)."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync

    def run(graph):
        step, opt, batches = _setup(golden, GradSync(2, chunk_mb=0.01))
        step.exact_graph_ln = True
        step.one_graph_exchange = True
        if graph:
            step.capture(batches, warmup=1)
            assert step._graph_has_exchange and not getattr(step, "capture_notes", [])
            for _ in range(2):
                step.replay()
        else:
            for _ in range(3):
                step.step(batches)
        torch.cuda.synchronize()
        assert opt.step_count == 3 and not ops.graph_ln_exchange_on()
        return opt.flat_p.clone()
    with ops.compute_mode("f32"):
        assert torch.equal(run(True), run(False))


def test_headwise_backward_equals_the_one_call_backward(golden):
    """Every head's backward as its own backward() call inside the head's stream context (so that the captured branches
    overlap), then the backbone from the three feature gradients: the parameters of one backward() over the whole
    objective, BIT FOR BIT -- eager and captured, with the heads on side streams and on the main stream."""
    from egopack_amd import ops

    def run(headwise, graph, parallel, one_pass=False):
        step, opt, batches = _setup(golden, None)
        step.headwise_backward, step.parallel_heads = headwise, parallel
        # (the one-pass PNR head needs the backward seed the headwise structure announces: with it on, the two structures
        #  run different arithmetic for that head -- a row reduction against an MFMA K walk -- and agree to rounding only)
        step.one_pass_heads = one_pass
        if graph:
            step.capture(batches, warmup=1)
            for _ in range(2):
                step.replay()
        else:
            for _ in range(3):
                step.step(batches)
        torch.cuda.synchronize()
        assert opt.step_count == 3
        return opt.flat_p.clone()
    for mode in ("f32", "bf16"):
        with ops.compute_mode(mode):
            ref = run(False, False, False)
            for headwise, graph, parallel in [(True, False, False), (True, False, True), (True, True, True), (False, True, True)]:
                assert torch.equal(run(headwise, graph, parallel), ref), (mode, headwise, graph, parallel)
            # with the one-pass head: captured == eager bit for bit, and within rounding of the contraction path
            eager = run(True, False, True, one_pass=True)
            assert torch.equal(run(True, True, True, one_pass=True), eager), mode
            close = (eager - ref).abs() <= (2e-4 if mode == "f32" else 2.1e-3)  # 3 Adam steps of lr 1e-3 on near-zero gradients
            assert close.double().mean() >= 0.999, (mode, float((eager - ref).abs().max()))
