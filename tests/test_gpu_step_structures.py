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

