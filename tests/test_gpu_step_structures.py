"""Execution STRUCTURES of the one-rank step that must not change its arithmetic (no process group involved: in-process)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_headwise_backward_equals_the_one_call_backward():
    """Every head's backward as its own backward() call inside the head's stream context (so that the captured branches
    overlap), then the backbone from the three feature gradients: the parameters of one backward() over the whole
    objective, BIT FOR BIT -- eager and captured, with the heads on side streams and on the main stream."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dist_child import setup_step
    from egopack_amd import ops

    def run(headwise, graph, parallel, one_pass=False):
        step, opt, batches = setup_step(None)
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


@pytest.mark.parametrize("streams,event_nodes", [(1, False), (2, False), (4, False), (2, True), (4, True)])
def test_segmented_replay_equals_the_runtime_replay(streams, event_nodes):
    """The captured step replayed as a plan of single-stream graphs (egk_graph_plan_*: one graph per fork-free path, events on
    the edges that cross streams) against the HIP runtime's own replay of the same capture: the same nodes, arguments and
    edges, so the parameters after seven steps (six replays back to back) are equal BIT FOR BIT, in both compute modes; the plan
    keeps every node.  ``event_nodes``: the cross-stream edges as event-record / event-wait nodes inside one graph per run of
    paths on a stream."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from dist_child import setup_step
    from egopack_amd import ops

    def run(segmented):
        step, opt, batches = setup_step(None)
        step.segmented_replay, step.segmented_event_nodes = segmented, event_nodes
        step.capture(batches, warmup=1)
        for _ in range(2 if not segmented else 6):  # (back to back: a replay's event waits must bind to THIS replay's records)
            step.replay()
        torch.cuda.synchronize()
        info = step._graph.info() if segmented else None
        return opt.flat_p.clone(), opt.flat_m.clone(), info
    for mode in ("f32", "bf16"):
        with ops.compute_mode(mode):
            p1, m1, info = run(streams)
            step0, opt0, batches0 = setup_step(None)
            step0.capture(batches0, warmup=1)
            for _ in range(6):
                step0.replay()
            torch.cuda.synchronize()
            p0, m0 = opt0.flat_p.clone(), opt0.flat_m.clone()
            assert torch.equal(p1, p0) and torch.equal(m1, m0), mode
            assert info["nodes"] >= 40 and 1 <= info["streams"] <= streams and info["segments"] <= info["nodes"]
            assert (info["cross_edges"] == 0) == (info["streams"] == 1)
