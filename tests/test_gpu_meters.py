"""Validation meters and loops on the GPU: the two metric kernels against the CPU oracle, the
meters against the reference-generated fixture (tests/golden/meters.pt) and the validation loops against what the
reference's validate.py handed to ``meter.update`` on the same models and batches (tests/golden/validate.pt)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import meters as OM  # noqa: E402

DEV = "cuda"
TRN_CFG = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": 40}


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import egopack_amd.data as data
    import egopack_amd.meters as meters
    import egopack_amd.ops as ops
    import egopack_amd.validate as validate
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask

    class NS:
        pass
    ns = NS()
    ns.__dict__.update(locals())
    return ns


class _DS:
    label_names = ["verbs", "nouns"]
    class_labels = [[f"v{i}" for i in range(7)], [f"n{i}" for i in range(11)]]


@pytest.mark.parametrize("rows,C", [(1, 2), (37, 7), (300, 478), (513, 115), (64, 1000)])
def test_label_rank_kernel_equals_oracle(M, rows, C):
    g = torch.Generator().manual_seed(rows + C)
    scores = (torch.randint(-4, 5, (rows, C), generator=g).float() / 2)  # few distinct values: ties
    labels = torch.randint(-1, C, (rows,), generator=g)
    scores[0, 0] = float("nan")
    pad = torch.full((rows, C + 3), 99.0)  # a padded row stride (logits are stored with ld % 8 == 0)
    pad[:, :C] = scores
    rank = M.meters.label_rank(pad.to(DEV)[:, :C], labels.to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(rank, OM.label_rank(scores.numpy(), labels.numpy()))


@pytest.mark.parametrize("N,Z,K", [(1, 1, 1), (5, 20, 5), (64, 20, 5), (3, 64, 2), (7, 0, 3)])
def test_edit_distance_kernel_equals_oracle(M, N, Z, K):
    g = torch.Generator().manual_seed(N * 100 + Z)
    pred = torch.randint(0, 4, (N, Z, K), generator=g)
    label = torch.randint(0, 4, (N, Z), generator=g)
    if Z:
        pred[0, :, 0] = label[0]
    out = M.meters.edit_distances(pred.to(DEV), label.to(DEV)).cpu()
    ref = torch.tensor([[OM.levenshtein(pred[n, :, k].tolist(), label[n].tolist()) for k in range(K)] for n in range(N)],
                       dtype=torch.int32).reshape(N, K)
    assert torch.equal(out, ref)


def test_recognition_meter_matches_reference_topk_functions(M, golden):
    g = golden("meters")["topk"]
    scores, labels = g["scores"], g["labels"]
    noun_scores = torch.randn(scores.shape[0], 11)
    y = torch.stack([labels, torch.full_like(labels, -1)], 1)  # nouns all ignored
    class DS11(_DS):  # the fixture's score matrix has 11 classes
        class_labels = [[f"v{i}" for i in range(11)], [f"n{i}" for i in range(11)]]
    m = M.meters.RecognitionMeter(DS11(), device=DEV)
    half = scores.shape[0] // 2
    for sl in (slice(0, half), slice(half, None)):  # two updates accumulate
        m.update((scores[sl].to(DEV), noun_scores[sl].to(DEV)), y[sl].to(DEV), torch.tensor(0.25, device=DEV))
    logs = m.get_logs()
    for k, acc in zip(g["ks"], g["accuracy"]):
        assert logs[f"verbs_top{k}"] == pytest.approx(acc, abs=1e-12)
    assert logs["verbs_mc"] == pytest.approx(g["recall"][1], abs=1e-12)
    assert m.verbs.mean_class(5) == pytest.approx(g["recall"][5], abs=1e-12)
    assert float(logs["verbs_class_acc"]["top-1"][3]) == pytest.approx(g["class3"][0], abs=1e-12)
    assert logs["nouns_top1"] == 0.0 and logs["loss"] == pytest.approx(0.25) and m.counter == scores.shape[0]
    # calibration error / "Brier score" (ego4d.py:52-53, :66-67) against the restated torchmetrics definition
    assert logs["verbs_calibration_erorr"] == pytest.approx(OM.multiclass_calibration_error(scores.numpy(), labels.numpy()), abs=1e-6)
    assert logs["verbs_brier_score"] == pytest.approx(OM.multiclass_calibration_error(scores.numpy(), labels.numpy(), 1, "l2"), abs=1e-6)
    assert logs["nouns_calibration_erorr"] == 0.0 and logs["nouns_brier_score"] == 0.0  # every noun label ignored
    assert any(ln.startswith("Verbs Brier score") for ln in m.print_logs())


def test_pnr_meter_matches_reference_meter_and_oracle(M, golden):
    g = golden("meters")["pnr"]
    gen = torch.Generator().manual_seed(5)
    y = (torch.rand(g["logits"].shape[0], generator=gen) < 0.3).long()
    m = M.meters.PNRMeter(device=DEV)
    m.update(g["logits"].to(DEV), y.to(DEV), g["batch"].to(DEV), g["start_frame"], g["end_frame"], g["pnr_frame"],
             torch.tensor([0.5, 1.5], device=DEV))
    logs = m.get_logs()
    assert logs["localization_error"] == pytest.approx(float(g["loc_errors"].mean()), abs=1e-9)
    probs = torch.sigmoid(g["logits"]).numpy()
    s = OM.binary_stats(probs, y.numpy())
    assert logs["accuracy"] == pytest.approx(s["accuracy"]) and logs["recall"] == pytest.approx(s["recall"])
    assert logs["auroc"] == pytest.approx(OM.binary_auroc(probs, y.numpy()), abs=1e-12)
    assert logs["loss"] == pytest.approx(1.0)


def test_lta_meter_matches_reference_meter(M, golden):
    g = golden("meters")["lta"]
    gen = torch.Generator().manual_seed(6)
    logits = (torch.randn(g["labels"].shape[0], 7, generator=gen), torch.randn(g["labels"].shape[0], 11, generator=gen))
    m = M.meters.LTAMeter(_DS(), device=DEV)
    m.update(tuple(l.to(DEV) for l in logits), g["labels"].to(DEV), [p.to(DEV) for p in g["predictions"]], torch.tensor(0.1, device=DEV))
    dv, dn = m.last_distances
    np.testing.assert_allclose(dv.cpu().numpy(), g["verbs"].numpy(), atol=1e-12)
    np.testing.assert_allclose(dn.cpu().numpy(), g["nouns"].numpy(), atol=1e-12)
    logs = m.get_logs()
    assert logs["verbs_ed"] == pytest.approx(float(g["verbs"].mean()), abs=1e-12)
    assert logs["verbs_top1"] == pytest.approx(OM.multiclass_accuracy(logits[0].numpy(), g["labels"][:, 0].numpy(), 1))


def _to_data(M, d):
    b = M.data.Data(**{k: v for k, v in d.items()})
    n = b.x.shape[0]
    b.graph = M.data.build_csr(b.edge_index, n)
    counts = torch.bincount(b.batch, minlength=b.num_graphs)
    b.ptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(counts, 0)])
    b.ptr32 = b.ptr.to(torch.int32)
    return b


class _Recorder:
    def __init__(self):
        self.calls = []

    def update(self, *args):
        self.calls.append(args)


@pytest.mark.parametrize("kind", ["ar", "oscc", "pnr", "lta"])
def test_validate_loops_feed_the_meter_what_the_reference_does(M, golden, kind):
    G = golden("validate")
    F_IN, S, H, HP, HEADS = G["dims"]
    model = M.Graph(F_IN, hidden_size=H, depth=2, temporal_pooling=TRN_CFG, num_segments=S)
    model.load_state_dict(G["sd"]["model"])
    task = {"ar": lambda: M.RecognitionTask(H, H, HEADS), "lta": lambda: M.LTATask(H, H, HEADS), "pnr": lambda: M.PNRTask(H, H),
            "oscc": lambda: M.OSCCTask(H, H)}[kind]()
    task.load_state_dict(G["sd"][kind])
    model, task = model.to(DEV), task.to(DEV)
    batches = [_to_data(M, d) for d in G[kind]["batches"]]
    rec = _Recorder()
    tol = dict(rtol=2e-4, atol=2e-4)
    with M.ops.compute_mode("f32"):
        if kind in ("ar", "oscc"):
            M.validate.validate(0, model, batches, rec, task, device=DEV)
        elif kind == "pnr":
            M.validate.validate_pnr(model, batches, rec, task, device=DEV)
        else:
            M.validate.validate_lta(model, batches, rec, task, device=DEV)
    assert len(rec.calls) == len(G[kind]["calls"])
    assert not model.training and not task.training
    for got, ref in zip(rec.calls, G[kind]["calls"]):
        logits, ref_logits = got[0], ref[0]
        if isinstance(ref_logits, (list, tuple)):
            for a, b in zip(logits, ref_logits):
                torch.testing.assert_close(a.float().cpu(), b, **tol)
        else:
            torch.testing.assert_close(logits.float().cpu().reshape(b_shape := ref_logits.shape), ref_logits, **tol)
        assert torch.equal(got[1].cpu(), ref[1])  # labels
        if kind in ("ar", "oscc"):
            torch.testing.assert_close(got[2].float().cpu(), ref[2], **tol)  # mean loss
            torch.testing.assert_close(got[3].float().cpu(), ref[3], **tol)  # pre-features (segment mean of x)
            torch.testing.assert_close(got[4].float().cpu(), ref[4], **tol)  # post-features
        elif kind == "pnr":
            torch.testing.assert_close(got[6].float().cpu(), ref[6], **tol)  # per-node BCE loss vector
            for i in (3, 4, 5):
                assert torch.equal(got[i].cpu(), ref[i])
        else:
            preds = got[2]
            assert len(preds) == 2 and all(p.shape == r.shape and p.dtype == torch.int64 for p, r in zip(preds, ref[2]))
            assert all(int(p.min()) >= 0 and int(p.max()) < c for p, c in zip(preds, HEADS))  # samples are class ids
            torch.testing.assert_close(got[3].float().cpu(), ref[3], **tol)


def test_lta_sampling_follows_the_logits(M):
    """generate_from_logits draws K categorical samples per node: with one dominant class they must all be it."""
    task = M.LTATask(32, 32, (7, 11)).to(DEV)
    lv = torch.full((44, 7), -30.0, device=DEV)
    lv[:, 3] = 30.0
    ln = torch.full((44, 11), -30.0, device=DEV)
    ln[:, 9] = 30.0
    preds, _ = task.generate_from_logits((lv, ln))
    assert preds[0].shape == (44, 5) and bool((preds[0] == 3).all()) and bool((preds[1] == 9).all())


def test_validate_metrics_on_synthetic_loaders(M):
    """The per-task validation pass of main_temporal on small synthetic loaders: every meter reports finite values in
    range, PNR frames ride through the collation, LTA sequences use the dataset's node count."""
    import main_temporal
    H = 64
    ds = {t: M.data.SyntheticTaskDataset(t, 12, 8, 3, 48, (7, 11), k=1, seed=3) for t in ("ar", "lta", "oscc", "pnr")}
    dl = {t: M.data.build_dataloader(d, 4, False, 0, False, 1) for t, d in ds.items()}
    b = next(iter(dl["pnr"]))
    assert b.start_frame.shape == (4,) and bool((b.pnr_frame >= b.start_frame).all()) and bool((b.pnr_frame <= b.end_frame).all())
    torch.manual_seed(0)
    model = M.Graph(48, hidden_size=H, depth=2, temporal_pooling={**TRN_CFG, "hidden_size": H}, num_segments=3).to(DEV)
    tasks = {"ar": M.RecognitionTask(H, H, (7, 11)).to(DEV), "lta": M.LTATask(H, H, (7, 11)).to(DEV),
             "oscc": M.OSCCTask(H, H).to(DEV), "pnr": M.PNRTask(H, H).to(DEV)}
    out = main_temporal.validate_metrics(1, model, tasks, ["ar", "lta", "oscc", "pnr"], ds, dl, DEV)
    assert set(out) == {"ar", "lta", "oscc", "pnr"}
    for t, logs in out.items():
        assert all(v == v and abs(v) < 1e6 for v in logs.values()), (t, logs)
    assert 0 <= out["ar"]["verbs_top1"] <= out["ar"]["verbs_top5"] <= 1
    assert 0 <= out["oscc"]["accuracy"] <= 1 and 0 <= out["pnr"]["auroc"] <= 1 and out["pnr"]["localization_error"] >= 0
    assert 0 <= out["lta"]["verbs_ed"] <= 1 and 0 <= out["lta"]["nouns_ed"] <= 1


def test_batch_sharded_validation_merges_to_the_single_pass_numbers(M):
    """SURVEY 8(e) caveat 5: two ranks validating alternate batches of the split and adding their meters report what one
    pass over the split reports -- counts exactly; the graph-LayerNorm statistics span a batch, which is why the shards
    are whole batches."""
    import main_temporal
    H = 64
    ds = {t: M.data.SyntheticTaskDataset(t, 22, 8, 3, 48, (7, 11), k=1, seed=5) for t in ("ar", "oscc", "pnr", "lta")}
    torch.manual_seed(1)
    model = M.Graph(48, hidden_size=H, depth=2, temporal_pooling={**TRN_CFG, "hidden_size": H}, num_segments=3).to(DEV)
    tasks = {"ar": M.RecognitionTask(H, H, (7, 11)).to(DEV), "lta": M.LTATask(H, H, (7, 11)).to(DEV),
             "oscc": M.OSCCTask(H, H).to(DEV), "pnr": M.PNRTask(H, H).to(DEV)}

    def run(t, rank, world):
        dl = M.data.build_dataloader(ds[t], 4, False, 0, False, 1, rank=rank, world_size=world, shard="batches")
        meter = M.meters.build_meter_for_dataset(ds[t], device=DEV)
        if t == "lta":
            M.validate.validate_lta(model, dl, meter, tasks[t], device=DEV)
        elif t == "pnr":
            M.validate.validate_pnr(model, dl, meter, tasks[t], device=DEV)
        else:
            M.validate.validate(1, model, dl, meter, tasks[t], device=DEV)
        return meter

    for t in ("ar", "oscc", "pnr", "lta"):
        whole = run(t, 0, 1)
        merged = run(t, 0, 2).merge(run(t, 1, 2))
        assert merged.counter == whole.counter and merged.loss_n == whole.loss_n
        for a, b in zip(merged._sums(), whole._sums()):
            if t == "lta" and a.dtype == torch.float64:
                continue  # the K=5 futures are sampled (multinomial): distances and nothing else depend on the draw order
            if a.dtype == torch.int64:
                assert torch.equal(a, b), t
            else:
                torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
        if t != "lta":
            got, want = merged.get_logs(), whole.get_logs()
            for k, v in want.items():
                if isinstance(v, (int, float)):
                    assert got[k] == pytest.approx(v, rel=1e-12, abs=1e-12), (t, k)
    assert run("ar", 0, 1).all_reduce().counter == 22 * 8  # no process group: a no-op (counter = label rows seen)


def test_batch_sharded_bank_pass_sums_to_the_single_pass_banks(M):
    """SURVEY 8(e) caveat 3: fp64 partial banks of alternate batches add up to the single-pass banks."""
    from egopack_amd import graphone as G1
    H = 64
    ds = M.data.SyntheticTaskDataset("ar", 40, 8, 3, 48, (5, 6), k=1, seed=9)
    torch.manual_seed(2)
    model = M.Graph(48, hidden_size=H, depth=2, temporal_pooling={**TRN_CFG, "hidden_size": H}, num_segments=3).to(DEV)
    ar, pnr = M.RecognitionTask(H, H, (5, 6)).to(DEV), M.PNRTask(H, H).to(DEV)
    mk = lambda r, w: M.data.build_dataloader(ds, 8, False, 0, True, 1, rank=r, world_size=w, shard="batches")
    whole = G1.build_graphone(model, ar, [ar, pnr], mk(0, 1), device=DEV)
    b0, c0 = G1.accumulate_banks(model, ar, [ar, pnr], mk(0, 2), device=DEV)
    b1, c1 = G1.accumulate_banks(model, ar, [ar, pnr], mk(1, 2), device=DEV)
    assert int(c0.sum()) + int(c1.sum()) == 2 * 40 and int(c0.sum()) == 2 * 24  # 5 batches: 3 + 2; count once per task
    merged = G1.finalise_banks({k: b0[k] + b1[k] for k in b0}, c0 + c1)
    assert merged.keys() == whole.keys()
    for k in whole:
        assert merged[k].shape == whole[k].shape
        torch.testing.assert_close(merged[k], whole[k], rtol=1e-6, atol=1e-7)


def test_validation_with_graphone_runs_the_backbone_once_in_bf16_mode(M):
    """validation with a GraphONE in bf16 mode ran the backbone twice (precise pass + bf16 pass).  Now the
    bf16 features are the roundings of the precise pass's taped results (validate.ONE_PASS, ops.dual_record / dual_replay):
    same neighbour lists (they come from the precise pass either way), logits within bf16 rounding of the two-pass ones and
    no further from the f32 run than those, and the backbone's contractions are launched once."""
    from egopack_amd.models.graphONE.graphONE import GraphONE
    H, K = 128, 256
    torch.manual_seed(3)
    ds = M.data.SyntheticTaskDataset("oscc", 16, 8, 3, 64, (7, 11), k=1, seed=4)
    model = M.Graph(64, hidden_size=H, depth=2, temporal_pooling={**TRN_CFG, "hidden_size": H}, num_segments=3).to(DEV)
    aux = ("ar", "lta", "pnr")
    primary = M.OSCCTask(H, H, aux_tasks=aux, average_logits=True).to(DEV)
    others = [M.RecognitionTask(H, H, (7, 11)).to(DEV), M.LTATask(H, H, (7, 11)).to(DEV), M.PNRTask(H, H).to(DEV)]
    for t, n in zip(others, aux):
        t.name = n
    gen = torch.Generator(device=DEV).manual_seed(5)
    banks = {n: torch.randn(K, H, device=DEV, generator=gen) for n in aux}
    g1 = GraphONE(banks, features_size=H, hidden_size=H, k=4, depth=2, residual=True).to(DEV)

    def run(mode, one_pass):
        dl = M.data.build_dataloader(ds, 8, False, 0, False, 1)
        rec = _Recorder()
        prev, M.validate.ONE_PASS = M.validate.ONE_PASS, one_pass
        try:
            with M.ops.compute_mode(mode):
                batches = []
                for b in dl:
                    b = b.to(DEV)
                    if mode == "bf16":
                        b.x = b.x.to(torch.bfloat16)
                    batches.append(b)
                M.ops.prof_reset()
                M.ops.prof_enable(True)
                M.validate.validate(0, model, batches, rec, primary, other_tasks=others, graphone=g1, late_fusion=True, device=DEV)
                torch.cuda.synchronize()
                M.ops.prof_enable(False)
                launches = sum(v["launches"] for k, v in M.ops.prof_report().items() if k.startswith("gemm"))
                M.ops.prof_reset()
        finally:
            M.validate.ONE_PASS = prev
        return [c[0].float().cpu() for c in rec.calls], [c[4].float().cpu() for c in rec.calls], launches

    f32, _, _ = run("f32", True)
    two, two_feat, n_two = run("bf16", False)
    one, one_feat, n_one = run("bf16", True)
    assert n_one < n_two, (n_one, n_two)  # the bf16 backbone pass launches no contraction of its own
    for a, b, r in zip(one, two, f32):
        torch.testing.assert_close(a, b, rtol=0, atol=6e-2)
        assert float((a - r).abs().max()) <= float((b - r).abs().max()) * 1.5 + 1e-2
    for a, b in zip(one_feat, two_feat):  # post-features [N, 1 + aux, H]: the aux parts come from the precise pass in both
        assert float((a - b).norm() / b.norm()) < 3e-2
