"""Child-process body of tests/test_gpu_dist.py: everything that needs a torch.distributed process group (an RCCL
communicator, its watchdog thread) or a hipGraph capture that holds collectives runs HERE, in a fresh process per test, never
inside the pytest process that carries the parity suite -- a process-fatal fault (SIGABRT out of the runtime, a watchdog
std::terminate) then fails ONE test instead of blanking the run (round-3 driver record: rc 134 in the third test file, 417
tests never reached).

    python tests/dist_child.py <scenario> <result.json>

Exit code 0 and ``{"ok": true, ...}`` in the result file on success; a Python exception writes ``{"ok": false, "error": ...}``
and exits 1; anything else (abort, kill) leaves no result and a negative / non-zero exit code for the parent to report.
The rendezvous port comes from the parent (EGK_TEST_PORT, a port it found free).  EGK_TEST_INJECT=abort makes the child call
os.abort() after its first collective: the harness's own test."""
import json
import os
import sys
import traceback
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GOLDEN = REPO / "tests" / "golden"


def golden(name):
    return torch.load(GOLDEN / f"{name}.pt", map_location="cpu", weights_only=False)


def setup_step(sync):
    """The reference's 4-task toy configuration (tests/golden/mtl_train.pt) as an MTLStep on cuda:0."""
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


# ---- scenarios (each returns a JSON-able dict of what it measured; assertions raise) ---------------------------------------
def dp_step_with_bf16_compressed_allreduce():
    """A 1-rank RCCL group driven through dist.GradSync as if the world had 2 ranks (the all-reduce sums a single
    contribution), bf16 gradient compression, eager and hipGraph replay: cast -> all-reduce on the side stream -> Adam
    reading bf16 gradients -- the plumbing of the 8-GPU run, in both capture modes."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync
    out = {}
    with ops.compute_mode("f32"):
        step, opt, batches = setup_step(None)
        opt.grad_scale = 0.5
        for _ in range(3):
            step.step(batches)
        ref = opt.flat_p.clone()
        step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01, compress="bf16"))
        for _ in range(3):
            step.step(batches)
        torch.cuda.synchronize()
        assert opt.grad_scale == 0.5 and step.sync._g16 is not None and step.sync._g16.dtype == torch.bfloat16
        # Adam normalises the update; bf16 rounding of a gradient moves an update by << lr
        torch.testing.assert_close(opt.flat_p, ref, rtol=0, atol=2e-4)
        eager = opt.flat_p.clone()
        # the DEFAULT N-rank capture is the three staged graphs with eager collectives between the launches; ONE graph that
        # holds the collectives and the per-chunk Adam launches is opt-in (StepBase.one_graph_exchange)
        step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01, compress="bf16"))
        assert step.one_graph_exchange is False
        for one_graph in (False, True):
            step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01, compress="bf16"))
            step.one_graph_exchange = one_graph
            step.capture(batches, warmup=1)
            assert step._graph_has_exchange is one_graph and bool(step._fuse_adam) is one_graph
            assert isinstance(step._graph, list) is (not one_graph)
            for _ in range(2):
                step.replay()
            torch.cuda.synchronize()
            assert opt.step_count == 3
            torch.testing.assert_close(opt.flat_p, eager, rtol=0, atol=2e-6)
            out[f"one_graph={one_graph}"] = float((opt.flat_p - eager).abs().max())
    return out


def staged_backward_equals_the_one_piece_backward(with_sync: bool):
    """The three-stage backward (heads | SAGE stack | TRN, cut at detached leaves, region-wise exchange between the
    stages) produces the parameters of the one-piece backward BIT FOR BIT -- eager and as three captured graphs."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync

    def run(staged, graph, one_graph=False):
        sync = GradSync(2, chunk_mb=0.01, compress="bf16") if with_sync else None
        step, opt, batches = setup_step(sync)
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
    return {"with_sync": with_sync}


def exact_graph_ln_mode_is_captured_with_the_exchange():
    """exact_graph_ln (graph-LayerNorm statistics summed over the ranks, forward and backward): with the one-graph exchange
    requested on an RCCL group its six small collectives per step are captured with the rest of the N-rank step -- the
    replayed graph gives the parameters of the eagerly issued steps bit for bit (1-rank group driven as a world of 2)."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync

    def run(graph):
        step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01))
        step.exact_graph_ln = True
        step.one_graph_exchange = True
        if graph:
            step.capture(batches, warmup=1)
            assert step._graph_has_exchange
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
        # without the opt-in the mode steps eagerly (its collectives cannot sit between staged graph launches): capture refuses
        step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01))
        step.exact_graph_ln = True
        try:
            step.capture(batches, warmup=1)
        except RuntimeError as e:
            assert "exact_graph_ln" in str(e)
        else:
            raise AssertionError("capture() accepted exact_graph_ln without the one-graph exchange")
    return {}


def failed_exchange_capture_is_not_retried():
    """A capture that holds collectives and fails must END the attempt in this process: the exception reaches the caller (no
    catch-and-continue onto the staged graphs on a communicator a failed capture may have invalidated)."""
    from egopack_amd import ops
    from egopack_amd.dist import GradSync
    with ops.compute_mode("f32"):
        step, opt, batches = setup_step(GradSync(2, chunk_mb=0.01))
        step.one_graph_exchange = True
        calls = []
        orig = step.sync.start

        def failing_start(*a, **k):
            calls.append(1)
            if len(calls) == 2:  # (after the first region's collectives went into the capture)
                raise RuntimeError("injected: the exchange failed inside the capture")
            return orig(*a, **k)
        step.capture(batches, warmup=1)  # (builds the flat buffers; a healthy capture first)
        step.sync.start = failing_start
        try:
            step.capture(batches, warmup=0)
        except RuntimeError as e:
            # (ending the capture of a graph whose forked streams were never joined raises hipErrorStreamCaptureUnjoined on
            #  top of the injected error: either way the caller sees an exception, with the cause in its chain)
            chain, seen = [], e
            while seen is not None:
                chain.append(str(seen))
                seen = seen.__cause__ or seen.__context__
            assert any("injected" in m for m in chain), chain
        else:
            raise AssertionError("the failed capture was swallowed")
        assert not isinstance(step._graph, list), "fell back to the staged graphs inside the same process"
    return {"start_calls": len(calls)}


def adam_reads_bf16_gradients():
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
    return {}


def main_temporal_metrics():
    """main_temporal.main(<EGK_TEST_ARGS, a JSON list>) with the exchange dry run (the process group of THIS child is the 1-rank
    RCCL group it steps over): the validation metrics of the trained model (tests/test_gpu_metric_target.py)."""
    import main_temporal
    args = json.loads(os.environ["EGK_TEST_ARGS"])
    torch.manual_seed(int(os.environ.get("EGK_TEST_SEED", "3")))
    out = main_temporal.main(args)
    return {"metrics": {f"{t}/{k}": float(v) for t, m in out["metrics"].items() for k, v in m.items()}}


SCENARIOS = {
    "main_temporal_metrics": main_temporal_metrics,
    "dp_step": dp_step_with_bf16_compressed_allreduce,
    "staged_nosync": lambda: staged_backward_equals_the_one_piece_backward(False),
    "staged_sync": lambda: staged_backward_equals_the_one_piece_backward(True),
    "exact_graph_ln": exact_graph_ln_mode_is_captured_with_the_exchange,
    "failed_capture": failed_exchange_capture_is_not_retried,
    "adam_bf16_grads": adam_reads_bf16_gradients,
    "noop": lambda: {},
}


def main():
    import faulthandler
    faulthandler.enable(all_threads=True)
    scenario, out_path = sys.argv[1], Path(sys.argv[2])
    result = {"ok": False, "scenario": scenario}
    try:
        if not torch.cuda.is_available():
            raise RuntimeError("needs a GPU")
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = os.environ["EGK_TEST_PORT"]
        dist.init_process_group("nccl", rank=0, world_size=1)
        t = torch.ones(8, device="cuda")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        if os.environ.get("EGK_TEST_INJECT") == "abort":
            os.abort()
        result.update(SCENARIOS[scenario]() or {})
        torch.cuda.synchronize()
        dist.destroy_process_group()
        result["ok"] = True
    except BaseException as e:  # noqa: BLE001
        result["error"] = f"{e!r}\n{traceback.format_exc()}"
    out_path.write_text(json.dumps(result))
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0 if result["ok"] else 1)  # (no interpreter teardown races with the process group's threads)


if __name__ == "__main__":
    main()
