"""Properties of the whole step at the benchmark model size (H = 1024, 1536-d x 3 features, real class counts),
where a CPU reference would take minutes: reproducibility, objective composition, finiteness, gradient flow."""
import argparse

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(**kw):
    a = argparse.Namespace(batch=16, T=32, hidden=1024, trn_hidden=1024, dropout=0.5, compute="bf16", workload="mtl",
                           bank=512, graphone_k=4, graphone_depth=2)
    a.__dict__.update(kw)
    return a


def _build(args):
    import bench
    from egopack_amd import engine, ops
    from egopack_amd.optim import FlatAdam
    ops.set_compute(args.compute)
    ops.manual_seed(99)
    model, tasks, crit, weights, dev, merged = bench.build_workload(args, 0, torch.device("cuda"))
    model.cuda().train()
    for t in tasks.values():
        t.cuda().train()
    params = [*model.parameters(), *(p for t in tasks.values() for p in t.parameters())]
    opt = FlatAdam(params, lr=1e-4, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
    return step, opt, dev, merged


@pytest.mark.parametrize("dropout", [0.0, 0.5])
def test_full_size_step_is_bitwise_reproducible(dropout):
    """No atomics anywhere on the step: two runs from the same seeds give bit-identical parameters (also with
    dropout: Philox masks are a pure function of seed and offset, and with the task heads on parallel streams)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    outs = []
    for _ in range(2):
        step, opt, dev, merged = _build(_args(dropout=dropout))
        for _ in range(3):
            total, vectors = step.step(dev, merged)
        torch.cuda.synchronize()
        outs.append((opt.flat_p.clone(), total.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()


def test_full_size_objective_composition_and_gradient_flow():
    """total == sum_t w_t * mean(loss_t) recomputed in fp64 from the returned loss vectors; ignored AR nodes give
    exactly 0; every parameter of an enabled task and of the backbone receives a non-zero gradient, the disabled
    task (OSCC) none."""
    from egopack_amd import ops
    step, opt, dev, merged = _build(_args(dropout=0.0))
    total, vectors = step.forward_backward(dev, merged)
    torch.cuda.synchronize()
    ref = sum(step.weights[t] * vectors[t].double().mean() for t in vectors)
    assert abs(total.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    ar_y = dev["ar"].y
    assert torch.equal(vectors["ar"][ar_y[:, 0] == -1], torch.zeros(int((ar_y[:, 0] == -1).sum()), device="cuda"))
    assert (vectors["ar"][ar_y[:, 0] != -1] > 0).all() and vectors["ar"].numel() == 16 * 32
    lta_y = dev["lta"].y
    assert torch.equal(vectors["lta"][lta_y[:, 0] == -1], torch.zeros(int((lta_y[:, 0] == -1).sum()), device="cuda"))
    for name, p in step.model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, name
    for t in ("ar", "lta", "pnr"):
        for name, p in step.tasks[t].named_parameters():
            assert p.grad is not None and p.grad.abs().sum() > 0, f"{t}.{name}"
    assert all(p.grad is None for p in step.tasks["oscc"].parameters())


def test_full_size_loss_decreases_under_training():
    """40 Adam steps on one fixed synthetic batch (lr 1e-4): the objective drops clearly (the step really trains)."""
    step, opt, dev, merged = _build(_args(dropout=0.0))
    first = last = None
    for i in range(40):
        total, _ = step.step(dev, merged)
        if i == 0:
            first = total.item()
    last = total.item()
    assert last < 0.8 * first, (first, last)


def test_full_size_graphone_step_runs_and_freezes_banks():
    """EgoPack novel-task step (OSCC primary, AR/LTA/PNR banks of 512 prototypes, k=4, depth 2) at H = 1024."""
    import bench
    from egopack_amd import engine, ops
    from egopack_amd.models.graphONE.graphONE import GraphONE
    from egopack_amd.optim import FlatAdam
    args = _args(workload="egopack_oscc", dropout=0.0)
    ops.set_compute("bf16")
    model, tasks, crit, weights, dev, merged = bench.build_workload(args, 0, torch.device("cuda"))
    model.cuda()
    for t in tasks.values():
        t.cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    banks = {t: torch.randn(512, 1024, device="cuda", generator=g) for t in ("ar", "lta", "pnr")}
    gone = GraphONE(banks, features_size=1024, hidden_size=1024, k=4, depth=2, residual=True).cuda()
    before = {t: gone.embeddings[t].weight.clone() for t in banks}
    params = [*model.parameters(), *(p for t in tasks.values() for p in t.parameters()), *gone.parameters()]
    opt = FlatAdam(params, lr=1e-4, weight_decay=1e-5)
    step = engine.EgoPackStep(model, tasks, gone, weights, opt)
    l0 = None
    for i in range(8):
        total, vectors = step.step(dev)
        l0 = total.item() if l0 is None else l0
    assert torch.isfinite(total) and total.item() < l0
    for t in banks:
        assert torch.equal(gone.embeddings[t].weight, before[t])
    assert vectors["oscc"].shape == (16,)
    # aux heads of the primary task and the GraphONE stages are trained, the aux tasks' projections are not
    assert tasks["oscc"].aux_classifiers["ar"][1].weight.grad.abs().sum() > 0
    assert gone.conv_stages["pnr"][1].module_3.weight.grad.abs().sum() > 0
    assert tasks["ar"].net[1].weight.grad is None
