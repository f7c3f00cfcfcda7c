"""End-to-end run of the two entry points on the GPU with tiny synthetic datasets: multi-task pre-training
writes a checkpoint with the reference's key layout, the EgoPack phase resumes from it, builds the
prototype banks and trains the novel task."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_main_temporal_then_main_egopack(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import main_egopack
    import main_temporal
    common = ["k=1", "batch_size=4", "num_epochs=2", "synthetic_samples=16", "model.hidden_size=64",
              "model.temporal_pooling.hidden_size=64", "oscc_feat_size=64", f"checkpoint_dir={tmp_path}", "save_model=True",
              "compute=f32", "optimizer.lr=1e-3"]
    main_temporal.main(common + ["enabled_tasks=[ar,lta,pnr]"])
    ckpt_path = tmp_path / "MTL_ar-lta-pnr" / "checkpoint.pth"
    ckpt = torch.load(ckpt_path, weights_only=False)
    assert {"temporal_graph", "task/recognition", "task/oscc", "task/lta", "task/pnr", "epoch"} <= set(ckpt)
    assert "net.module_0.lin_l.weight" in ckpt["temporal_graph"] and ckpt["epoch"] == 2
    assert all(torch.isfinite(v).all() for v in ckpt["temporal_graph"].values())
    common[3] = "synthetic_samples=256"  # build_graphone reads the AR split with batch 256, drop_last=True
    main_egopack.main(common + ["enabled_tasks=[oscc]", "enable_graphone=True", f"resume_from={ckpt_path}", "graphone.k=4",
                                "graphone.depth=2", "graphone.residual=True", "graphone.hidden_size=64",
                                "+graphone.features_size=64", "artifact_prefix=EGO"])
    ego = torch.load(tmp_path / "EGO_egopack_oscc" / "checkpoint.pth", weights_only=False)
    assert "graphone" in ego and any(k.startswith("embeddings.ar") for k in ego["graphone"])
    assert any(k.startswith("conv_stages.pnr.1.module_3") for k in ego["graphone"])
    # the backbone moved (backprop_temporal_graph defaults to true), the frozen banks did not
    moved = sum((ego["temporal_graph"][k] - ckpt["temporal_graph"][k]).abs().sum() for k in ckpt["temporal_graph"])
    assert moved > 0


def test_flat_adam_state_dict_round_trip_and_torch_interchange():
    """FlatAdam.state_dict() has torch.optim.Adam's per-parameter layout: a torch Adam on copies of the parameters
    loads it and both continue identically; a FlatAdam that loads a state BEFORE its flat buffers exist applies it
    when the first step builds them."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(33, 16), (16,), (70, 8), (5,)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(5)]

    def run(opt, params, its):
        for it in its:
            for p, gr in zip(params, grads[it]):
                if p.grad is None:
                    p.grad = gr.clone().to(p.device)
                else:
                    p.grad.copy_(gr)
            opt.step()

    a = [p.clone().cuda().requires_grad_(True) for p in ps]
    opt_a = FlatAdam(a, lr=1e-2, weight_decay=1e-3)
    run(opt_a, a, range(3))
    sd = opt_a.state_dict()
    assert set(sd["state"]) == {0, 1, 2, 3} and sd["state"][0]["exp_avg"].shape == (33, 16) and float(sd["state"][2]["step"]) == 3
    # torch.optim.Adam continues from it
    t = [p.detach().clone().cpu().requires_grad_(True) for p in a]
    opt_t = torch.optim.Adam(t, lr=1e-2, weight_decay=1e-3)
    opt_t.load_state_dict({"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in sd["state"].items()},
                           "param_groups": sd["param_groups"]})
    # a fresh FlatAdam loads it before materialising
    b = [p.detach().clone().requires_grad_(True) for p in a]
    opt_b = FlatAdam(b, lr=1.0)  # wrong lr on purpose: the loaded param_groups must win
    opt_b.load_state_dict(sd)
    assert opt_b.step_count == 3 and opt_b.param_groups[0]["lr"] == 1e-2
    run(opt_a, a, range(3, 5))
    run(opt_t, t, range(3, 5))
    run(opt_b, b, range(3, 5))
    for pa, pt, pb in zip(a, t, b):
        assert torch.equal(pa.detach(), pb.detach())  # resumed == uninterrupted, bitwise
        torch.testing.assert_close(pa.detach().cpu(), pt.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.timeout(600)
def test_main_temporal_resume_equals_uninterrupted_run(tmp_path):
    """3 epochs in one go == 2 epochs, checkpoint (weights + Adam moments + schedule), resume, 1 more epoch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import main_temporal
    base = ["k=1", "batch_size=4", "synthetic_samples=16", "model.hidden_size=64", "model.temporal_pooling.hidden_size=64",
            "oscc_feat_size=64", "save_model=True", "compute=f32", "optimizer.lr=1e-3", "enabled_tasks=[ar,pnr]",
            "lr_scheduler.T_max=3", "use_graph=false"]  # (dropout stays on: its streams are part of the checkpoint; eager: a resumed
    # run would capture at another step than the uninterrupted one, and the replayed step draws its masks from other offsets)
    main_temporal.main(base + ["num_epochs=3", f"checkpoint_dir={tmp_path / 'full'}"])
    main_temporal.main(base + ["num_epochs=2", f"checkpoint_dir={tmp_path / 'part'}"])
    part = tmp_path / "part" / "MTL_ar-pnr" / "checkpoint.pth"
    ck = torch.load(part, weights_only=False)
    assert ck["epoch"] == 2 and "optimizer" in ck and "scheduler" in ck
    main_temporal.main(base + ["num_epochs=3", f"checkpoint_dir={tmp_path / 'resumed'}", f"resume_from={part}"])
    full = torch.load(tmp_path / "full" / "MTL_ar-pnr" / "checkpoint.pth", weights_only=False)
    res = torch.load(tmp_path / "resumed" / "MTL_ar-pnr" / "checkpoint.pth", weights_only=False)
    assert res["epoch"] == 3
    for key in ("temporal_graph", "task/recognition", "task/pnr"):
        for k, v in full[key].items():
            torch.testing.assert_close(res[key][k], v, rtol=0, atol=0, msg=lambda s: f"{key}.{k}: {s}")


def test_train_step_replays_the_captured_step_on_static_shape_batches():
    """StepBase.train_step (what the training loops call): eager for the first steps, then ONE capture, then value
    copies + replays for every batch with the capture's signature and the eager step for any other (a short last
    batch) -- the parameters of a plain eager loop over the same batches, bit for bit (no dropout: the eager and the
    replayed step draw their masks from different Philox offsets)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd import data as D
    from egopack_amd import engine, ops
    from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask
    from egopack_amd.optim import FlatAdam
    H, heads = 64, (7, 11)
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}

    class DS:
        has_joint_label, num_labels = False, 2

    def run(use_graph):
        torch.manual_seed(3)
        model = Graph(48, hidden_size=H, depth=2, temporal_pooling=trn, num_segments=3).cuda()
        tasks = {"ar": RecognitionTask(H, H, heads).cuda(), "oscc": OSCCTask(H, H).cuda(), "lta": LTATask(H, H, heads).cuda(),
                 "pnr": PNRTask(H, H).cuda()}
        crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), DS()), "lta": MetricSelectorWrapper(CrossEntropyNone(), DS()),
                "oscc": CrossEntropyNone(), "pnr": BCEWithLogitsNone()}
        live = [*model.parameters(), *(p for t in ("ar", "lta", "pnr") for p in tasks[t].parameters())]
        opt = FlatAdam(live, lr=1e-3, weight_decay=1e-5)
        weights = {"ar": 1.0, "lta": 0.5, "pnr": 2.0, "oscc": 0.0}
        step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
        step.use_graph = use_graph
        dsets = {t: D.SyntheticTaskDataset(t, 22, 8, 3, 48, heads, k=1, seed=11) for t in ("ar", "lta", "pnr")}
        loaders = {t: D.build_dataloader(d, 4, False, 0, False, 1) for t, d in dsets.items()}  # 5 full batches + one of 2
        losses = []
        with ops.compute_mode("f32"):
            for bs in zip(*(loaders[t] for t in ("ar", "lta", "pnr"))):
                host = dict(zip(("ar", "lta", "pnr"), bs))
                batches, merged = engine.stage_batches(host, "cuda", ("ar", "lta", "oscc", "pnr"))
                total, vectors = step.train_step(batches, merged)
                losses.append((float(total), {t: v.clone() for t, v in vectors.items()}))
        torch.cuda.synchronize()
        # the running per-task loss sums accumulate inside the step (eager and replayed alike): what a loop that sums the returned
        # vectors step by step would log
        sums = step.loss_sums()
        for t in ("ar", "lta", "pnr"):
            want = sum(float(v[t].double().sum()) for _, v in losses)
            assert sums[t][1] == sum(v[t].numel() for _, v in losses)
            assert abs(sums[t][0] - want) <= 1e-5 * max(abs(want), 1.0), (t, sums[t], want)
        assert "oscc" not in sums and step.loss_sums()["ar"] == (0.0, 0)  # (enabled tasks only; reading clears them)
        return opt.flat_p.clone(), losses, step

    p_eager, l_eager, _ = run(False)
    p_graph, l_graph, step = run(True)
    assert getattr(step, "_train_static", None) is not None and step._steps_seen == 6  # captured at step 3, replayed, then eager
    assert torch.equal(p_graph, p_eager)
    for (ta, va), (tb, vb) in zip(l_eager, l_graph):
        assert ta == tb
        for t in va:
            assert torch.equal(va[t], vb[t]), t


def test_arena_staging_with_the_staging_thread_trains_like_the_tensor_by_tensor_path():
    """The live loops' staging: resident datasets whose batches the native builder writes into ONE buffer each
    (data.Arena), the merged batch from one host call, one memcpy per batch into the transfer buffer, the feature block gathered
    from the device copy of the store rows, all of it on a thread of its own two steps ahead -- against the numpy builder with
    tensor-by-tensor merge / packing staged inline: the same parameters after the same steps, bit for bit, and a replayed step
    builds no tensor view on the host."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from egopack_amd import data as D
    from egopack_amd import engine, ops
    from egopack_amd import train as T
    from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask
    from egopack_amd.optim import FlatAdam
    H, heads, order = 64, (7, 11), ("ar", "lta", "oscc", "pnr")
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}

    class DS:
        has_joint_label, num_labels = False, 2

    def run(native, depth):
        torch.manual_seed(3)
        model = Graph(48, hidden_size=H, depth=2, temporal_pooling=trn, num_segments=3).cuda()
        tasks = {"ar": RecognitionTask(H, H, heads).cuda(), "oscc": OSCCTask(H, H).cuda(), "lta": LTATask(H, H, heads).cuda(),
                 "pnr": PNRTask(H, H).cuda()}
        crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), DS()), "lta": MetricSelectorWrapper(CrossEntropyNone(), DS()),
                "oscc": CrossEntropyNone(), "pnr": BCEWithLogitsNone()}
        live = [*model.parameters(), *(p for t in ("ar", "lta", "pnr") for p in tasks[t].parameters())]
        opt = FlatAdam(live, lr=1e-3, weight_decay=1e-5)
        weights = {"ar": 1.0, "lta": 0.5, "pnr": 2.0, "oscc": 0.0}
        step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True)
        dsets = {t: D.SyntheticResidentDataset(t, 48, 8, 3, 48, heads, k=1, seed=11, split="train", n_videos=3, frames=500)
                 for t in ("ar", "lta", "pnr")}
        for d in dsets.values():
            d.native_batches = native
        with ops.compute_mode("f32"):
            store = T.build_feature_store(dsets, "cuda")
            loaders = {t: D.build_dataloader(d, 4, True, 0, True, seed=5) for t, d in dsets.items()}
            hosts = (dict(zip(order, b)) for b in D.multiloader([loaders.get(t) for t in order], [weights[t] for t in order]))
            fills, n = [], 0
            for batches, merged in engine.StagedBatches(hosts, "cuda", order, fused=True, store=store, dtype=ops.act_dtype(), depth=depth,
                                                        step=step if native else None):
                before = D.LazyData.fills
                step.train_step(batches, merged)
                n += 1
                if n > step.graph_after + 2:
                    fills.append(D.LazyData.fills - before)
        torch.cuda.synchronize()
        return opt.flat_p.clone(), step, fills, n

    p_ref, step_ref, _, n_ref = run(False, 0)
    p_new, step_new, fills, n_new = run(True, 2)
    assert n_ref == n_new == 12 and step_new.loop_counts["replayed"] >= 8
    assert step_new.gathers_inputs and not step_ref.gathers_inputs  # (the last steps' feature blocks were gathered by the step itself)
    assert torch.equal(p_new, p_ref)
    if D.SyntheticResidentDataset.native_batches:
        assert fills and max(fills) == 0, fills  # (the staging thread may be mid-step: it builds no view either)


@pytest.mark.timeout(600)
def test_main_temporal_on_the_device_resident_feature_store(tmp_path):
    """dataset_*=synthetic_resident: the datasets deliver index matrices over ONE feature table in HBM; the training step
    gathers its rows on the device (stage_batches with a store), the captured step is replayed on them, validation goes
    through the resident adapter -- and the features are the reference's np.take of the same rows."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import main_temporal
    from egopack_amd import data as D
    from egopack_amd import train as T
    from egopack_amd.feature_store import FeatureStore
    groups = [f"{g}=synthetic_resident" for g in ("dataset_recognition", "dataset_lta", "dataset_oscc", "dataset_pnr")]
    args = [*groups, "k=1", "batch_size=4", "synthetic_samples=24", "model.hidden_size=64", "model.temporal_pooling.hidden_size=64",
            "oscc_feat_size=64", "num_epochs=2", "enabled_tasks=[ar,lta,pnr]", "save_model=True", f"checkpoint_dir={tmp_path}"]
    main_temporal.main(args)
    ck = torch.load(tmp_path / "MTL_ar-lta-pnr" / "checkpoint.pth", weights_only=False)
    assert ck["epoch"] == 2 and all(torch.isfinite(v).all() for v in ck["temporal_graph"].values() if v.is_floating_point())
    # what the step trains on = the reference's host-side np.take of the same rows
    cfg = T.load_config(args)
    ds = T.build_datasets(cfg, "train")
    store = T.build_feature_store(ds, "cuda")
    assert isinstance(store, FeatureStore)
    b = D.collate([ds["ar"][i] for i in range(3)])
    got = store.gather(b.x_idx.cuda(), dtype=torch.float32).cpu()
    table = torch.cat([torch.from_numpy(v) for v in ds["ar"].videos.values()])
    want = table[b.x_idx.clamp(min=0)] * (b.x_idx >= 0).unsqueeze(-1)
    torch.testing.assert_close(got, want.to(torch.bfloat16).float(), rtol=0, atol=0)


@pytest.mark.timeout(600)
def test_main_egopack_on_the_device_resident_feature_store(tmp_path):
    """Both phases over ONE feature table in HBM: main_temporal (dataset_*=synthetic_resident) writes the MTL checkpoint,
    main_egopack resumes from it -- its prototype-bank pass and its validation go through the resident adapter
    (train.ResidentLoader), its training step gathers the rows on the device (StagedBatches with a store) and replays the
    captured step."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import main_egopack
    import main_temporal
    groups = [f"{g}=synthetic_resident" for g in ("dataset_recognition", "dataset_lta", "dataset_oscc", "dataset_pnr")]
    common = [*groups, "k=1", "batch_size=4", "synthetic_samples=256", "model.hidden_size=64", "model.temporal_pooling.hidden_size=64",
              "oscc_feat_size=64", f"checkpoint_dir={tmp_path}", "save_model=True", "optimizer.lr=1e-3"]
    main_temporal.main(common + ["num_epochs=1", "enabled_tasks=[ar,lta,pnr]"])
    ckpt = tmp_path / "MTL_ar-lta-pnr" / "checkpoint.pth"
    main_egopack.main(common + ["num_epochs=2", "enabled_tasks=[oscc]", "enable_graphone=True", f"resume_from={ckpt}", "graphone.k=4",
                                "graphone.depth=2", "graphone.residual=True", "graphone.hidden_size=64", "+graphone.features_size=64",
                                "artifact_prefix=EGO"])
    ego = torch.load(tmp_path / "EGO_egopack_oscc" / "checkpoint.pth", weights_only=False)
    assert "graphone" in ego and all(torch.isfinite(v).all() for v in ego["graphone"].values() if v.is_floating_point())
    assert all(torch.isfinite(v).all() for v in ego["temporal_graph"].values() if v.is_floating_point())
