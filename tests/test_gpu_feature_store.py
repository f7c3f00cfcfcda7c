"""Resident input pipeline on the GPU: the gather kernel against numpy ``take``, and a training
step fed from the device-resident store against the same step fed the reference's way (features built on the host)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def X():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import egopack_amd.data as data
    import egopack_amd.engine as engine
    import egopack_amd.feature_store as fs
    import egopack_amd.ops as ops

    class NS:
        pass
    ns = NS()
    ns.__dict__.update(locals())
    return ns


@pytest.mark.parametrize("tdt,odt", [(torch.float32, torch.float32), (torch.float32, BF), (BF, BF), (BF, torch.float32)])
@pytest.mark.parametrize("rows,cols,n", [(50, 1536, 300), (7, 48, 33), (5, 13, 9), (3, 8, 0)])
def test_gather_rows_equals_numpy_take(X, tdt, odt, rows, cols, n):
    g = torch.Generator().manual_seed(rows + cols + n)
    table = torch.randn(rows, cols, generator=g)
    idx = torch.randint(-2, rows + 2, (n,), generator=g)  # includes -1/-2 (zero clip) and out-of-range rows
    store = X.fs.FeatureStore({"v": table.numpy()}, device=DEV, dtype=tdt)
    out = store.gather(idx.view(-1, 1) if n else idx.view(0, 1), dtype=odt)
    assert out.shape == (n, 1, cols) and out.dtype == odt
    src = table.to(tdt)  # what the store holds
    ok = (idx >= 0) & (idx < rows)
    ref = torch.where(ok[:, None], src[idx.clamp(0, rows - 1)].to(odt), torch.zeros((), dtype=odt))
    assert torch.equal(out.cpu().view(n, cols), ref)


def test_store_layout_and_chunked_upload(X):
    vids = {"a": np.arange(30, dtype=np.float32).reshape(10, 3), "b": 100 + np.arange(21, dtype=np.float32).reshape(7, 3)}
    store = X.fs.FeatureStore(vids, device=DEV, dtype=torch.float32, chunk_rows=4)
    assert store.offsets == {"a": (0, 10), "b": (10, 7)} and store.rows == 17 and store.features_size == 3
    assert torch.equal(store.table.cpu(), torch.from_numpy(np.concatenate([vids["a"], vids["b"]])))
    with pytest.raises(ValueError):
        X.fs.FeatureStore({"a": vids["a"], "c": np.zeros((2, 4), np.float32)}, device=DEV)


def test_resident_batches_equal_host_built_batches_and_train_identically(X):
    """Same samples through both pipelines: x gathered in HBM == x taken on the host (bitwise), and one multi-task
    step from either gives the same loss."""
    from egopack_amd.criterion import BCEWithLogitsNone, MetricSelectorWrapper, CrossEntropyNone
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import PNRTask, RecognitionTask
    from egopack_amd.optim import FlatAdam
    H, F, S, T = 64, 48, 3, 6
    ds = {t: X.data.SyntheticResidentDataset(t, 8, T, S, F, (7, 11), k=1, seed=4, n_videos=3, frames=120) for t in ("ar", "pnr")}
    stores = {t: X.fs.FeatureStore(d.videos, device=DEV, dtype=torch.float32) for t, d in ds.items()}
    # one store for all tasks in a real run; here both datasets share the seed, hence the same videos
    assert torch.equal(stores["ar"].table, stores["pnr"].table)
    store = stores["ar"]
    host, res = {}, {}
    for t, d in ds.items():
        uid, starts, ends = d.windows[0]
        ends = ends.copy()
        ends[1] = starts[1]  # an empty action window: the reference's all-zero clip
        d.windows[0] = (uid, starts, ends)
        state = d.rng.get_state()
        res[t] = X.data.collate([d[i] for i in range(4)])
        d.rng.set_state(state)
        host[t] = X.data.collate([d.host_item(i) for i in range(4)])
    assert res["ar"].x is None and res["ar"].x_idx.shape == (4 * T, S) and bool((res["ar"].x_idx < 0).any())  # zero clips occur
    dev_res, merged_res = X.engine.stage_batches(res, DEV, ("ar", "pnr"), store=store, dtype=torch.float32)
    dev_host, merged_host = X.engine.stage_batches(host, DEV, ("ar", "pnr"))
    for t in ("ar", "pnr"):
        assert torch.equal(dev_res[t].x, dev_host[t].x)
    assert torch.equal(merged_res.x, merged_host.x) and torch.equal(merged_res.seg_ptr, merged_host.seg_ptr)

    def one_step(batches, merged):
        torch.manual_seed(0)
        trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
        model = Graph(F, hidden_size=H, depth=2, temporal_pooling=trn, num_segments=S).to(DEV)
        tasks = {"ar": RecognitionTask(H, H, (7, 11)).to(DEV), "pnr": PNRTask(H, H).to(DEV)}
        crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), ds["ar"]), "pnr": BCEWithLogitsNone()}
        params = [*model.parameters(), *tasks["ar"].parameters(), *tasks["pnr"].parameters()]
        step = X.engine.MTLStep(model, tasks, crit, {"ar": 1.0, "pnr": 1.0}, FlatAdam(params, lr=1e-3))
        with X.ops.compute_mode("f32"):
            total, _ = step.step(batches, merged)
        return float(total)
    assert one_step(dev_res, merged_res) == one_step(dev_host, merged_host)


def test_store_builds_the_reference_datasets_samples(X, golden):
    """The whole input pipeline against the reference datasets' ``get`` (tests/golden/pipeline.pt): index matrices from the
    host builders, feature blocks from the resident store on the device (plain and interpolating gather) -- x bit for bit
    for AR, LTA ('avg' / 'zero' forecast nodes), OSCC and PNR samples, incl. all-zero clips."""
    import sys
    sys.path.insert(0, "tests")
    from test_feature_store_cpu import _build, _first_rows
    G = golden("pipeline")
    first, table = _first_rows(G)
    lens = {k: v.shape[0] for k, v in G["videos"].items()}
    store = X.fs.FeatureStore({k: v.numpy() for k, v in G["videos"].items()}, device=DEV, dtype=torch.float32)
    assert {k: v[0] for k, v in store.offsets.items()} == first
    for case in G["cases"]:
        item = _build(case, first, lens, G["stride"])
        x = store.gather_item(item, dtype=torch.float32)
        assert torch.equal(x.cpu(), case["data"]["x"]), (case["kind"], case["split"])
    # bf16 output of the interpolating form: the f32 result rounded once
    case = next(c for c in G["cases"] if c["kind"] == "pnr")
    item = _build(case, first, lens, G["stride"])
    assert torch.equal(store.gather_item(item, dtype=BF).cpu(), case["data"]["x"].to(BF))
