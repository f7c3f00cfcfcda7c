"""Host half of the resident input pipeline: the segment-sampling index arithmetic against the
reference's own functions run on seeded numpy state (tests/golden/sampling.pt, oracle/make_golden_sampling.py)."""
import numpy as np
import pytest
import torch

from egopack_amd import feature_store as FS


def test_sampling_indices_match_reference_and_consume_the_same_random_numbers(golden):
    for c in golden("sampling")["index_cases"]:
        np.random.set_state(c["state_before"])
        rnd = FS.random_sampling_indices(c["size"], c["n"])
        assert rnd.tolist() == c["random"], c
        assert int(np.random.get_state()[2]) == c["state_after_pos"]  # same position in the Mersenne stream afterwards
        assert FS.uniform_sampling_indices(c["size"], c["n"]).tolist() == c["uniform"], c


def test_window_rows_reproduce_reference_takes_including_zero_clips(golden):
    G = golden("sampling")
    feats = G["features"].numpy()
    first = 7  # the video sits at row 7 of a larger store
    store = np.concatenate([np.full((first, feats.shape[1]), -5.0, np.float32), feats, np.full((3, feats.shape[1]), -9.0, np.float32)])
    for t in G["takes"]:
        np.random.set_state(t["state_before"])
        rows = FS.window_rows(first, feats.shape[0], t["a"], t["b"], t["n"], t["random"])
        got = np.where(rows[:, None] >= 0, store[np.maximum(rows, 0)], 0.0)
        np.testing.assert_array_equal(got, t["out"].numpy(), err_msg=str((t["a"], t["b"], t["n"], t["random"])))


def test_private_generator_gives_the_same_stream_as_the_global_one():
    np.random.seed(5)
    a = FS.random_sampling_indices(50, 3)
    b = FS.random_sampling_indices(50, 3, np.random.RandomState(5))
    assert a.tolist() == b.tolist()


def _first_rows(G):
    first, off = {}, 0
    for uid, arr in G["videos"].items():
        first[uid] = off
        off += arr.shape[0]
    return first, np.concatenate([v.numpy() for v in G["videos"].values()])


def _build(case, first, lens, stride):
    import random
    e = case["entry"]
    np.random.set_state(case["np_state"])
    random.setstate(case["py_state"])
    train = case["split"] == "train"
    if case["kind"] == "ar":
        return FS.ar_item(first[e["video"]], lens[e["video"]], e["actions"], e["window_size"], stride, 3,
                          train and e["randomize_train"])
    if case["kind"] == "lta":
        return FS.lta_item(first[e["video"]], lens[e["video"]], e["input"], e["forecast_labels"], 4, stride, 3, train, e["append_node"])
    v = e["video_uid"]
    if case["kind"] == "oscc":
        return FS.oscc_item(first[v], lens[v], e["start_frame"], e["end_frame"], e["pnr_frame"], e["state_change"], stride, 3, train,
                            aug_prob=0.0)
    return FS.pnr_item(first[v], lens[v], e["start_frame"], e["end_frame"], e["pnr_frame"], e["start_sec"], e["end_sec"], stride, 16, train)


def test_sample_builders_reproduce_the_reference_datasets_get(golden):
    """ar_item / lta_item / oscc_item / pnr_item + the numpy evaluation of their index matrices against what the
    reference's Ego4dRecognitionDataset / Ego4dLTADataset / Ego4dOSCCDataset / Ego4dPNRDataset ``get`` returned for the
    same annotations and random states (tests/golden/pipeline.pt): features bit for bit, labels, positions, frames."""
    G = golden("pipeline")
    first, table = _first_rows(G)
    lens = {k: v.shape[0] for k, v in G["videos"].items()}
    seen = set()
    for case in G["cases"]:
        item = _build(case, first, lens, G["stride"])
        ref = case["data"]
        x = FS.take_reference(table, item)
        np.testing.assert_array_equal(x, ref["x"].numpy(), err_msg=f"{case['kind']} {case['split']} x")
        y_ref = ref["y"]
        if torch.is_tensor(y_ref):
            np.testing.assert_array_equal(item["y"], y_ref.numpy())
        else:
            assert item["y"] == y_ref
        np.testing.assert_array_equal(item["pos"], ref["pos"].numpy())
        if case["kind"] == "pnr":
            assert int(item["start_frame"]) == int(ref["start_frame"]) and int(item["end_frame"]) == int(ref["end_frame"])
            assert item["pnr_frame"] == ref["pnr_frame"]
        seen.add((case["kind"], case["split"]))
        if (item["lo"] < 0).any():
            seen.add("zero-clip")
    assert {("ar", "train"), ("ar", "val"), ("lta", "train"), ("oscc", "validation"), ("pnr", "train"), "zero-clip"} <= seen


# ---- whole-batch sampling on the same random stream (the live loaders' builders) ----------------------------------------
def _windows(W=700, seed=5):
    rs = np.random.RandomState(seed)
    vl = rs.randint(100, 4000, W)
    st = rs.randint(-5, 3900, W)
    en = st + rs.randint(0, 400, W)
    en[::53] = st[::53]  # empty windows
    st[::71] = vl[::71] + 3  # windows past the end of the video
    return rs.randint(0, 10 ** 6, W), vl, st, en


@pytest.mark.parametrize("random", [True, False])
@pytest.mark.parametrize("n", [1, 3, 8])
def test_window_rows_batch_equals_the_per_window_calls_and_leaves_the_stream_where_they_do(random, n):
    fr, vl, st, en = _windows()
    W = len(fr)
    a, b, c = (np.random.RandomState(7) for _ in range(3))
    for _ in range(2):  # (twice: the second call starts from the advanced generator state)
        ref = np.stack([FS.window_rows(fr[w], vl[w], st[w], en[w], n, random, a) for w in range(W)])
        np.testing.assert_array_equal(FS.window_rows_batch(fr, vl, st, en, n, random, b), ref)
        np.testing.assert_array_equal(FS.window_rows_batch(fr, vl, st, en, n, random, c, native=False), ref)
    assert (ref < 0).any() and (ref >= 0).any()
    assert a.randint(1 << 30) == b.randint(1 << 30) == c.randint(1 << 30)
    np.testing.assert_array_equal(a.standard_normal(4), b.standard_normal(4))  # (the cached-gaussian half of the state too)


def test_window_rows_batch_on_the_module_level_stream():
    fr, vl, st, en = _windows(300)
    for seed in (3, 4):
        np.random.seed(seed)
        ref = np.stack([FS.window_rows(fr[w], vl[w], st[w], en[w], 3, True) for w in range(len(fr))])
        x = np.random.rand()
        np.random.seed(seed)
        got = FS.window_rows_batch(fr, vl, st, en, 3, True)
        assert np.array_equal(ref, got) and x == np.random.rand()


def test_randint_sequence_replays_numpys_masked_rejection_sampling():
    highs = np.array([0, 1, 2, 3, 5, 17, 64, 65, 1000, 1 << 20, (1 << 32), 1, 7], dtype=np.int64)
    for n in (1, 3, 16):
        a, b = np.random.RandomState(11), np.random.RandomState(11)
        ref = np.stack([a.randint(h, size=n) if h > 0 else np.zeros(n, dtype=np.int64) for h in highs])
        np.testing.assert_array_equal(FS.randint_sequence(b, highs, n), ref)
        assert a.randint(1 << 30) == b.randint(1 << 30)
    with pytest.raises(ValueError):
        FS.randint_sequence(np.random.RandomState(1), np.array([(1 << 32) + 2]), 2)  # a bound numpy draws 64-bit words for


def test_generator_state_route_without_direct_access(monkeypatch):
    """A generator whose state block cannot be viewed in place goes through get_state / set_state with the same results."""
    fr, vl, st, en = _windows(200)

    class Wrapped:  # no ``_bit_generator``: only the public state interface
        def __init__(self, seed):
            self.rs = np.random.RandomState(seed)
        get_state = lambda self: self.rs.get_state()
        set_state = lambda self, s: self.rs.set_state(s)
    a, b = np.random.RandomState(9), Wrapped(9)
    np.testing.assert_array_equal(FS.window_rows_batch(fr, vl, st, en, 3, True, a), FS.window_rows_batch(fr, vl, st, en, 3, True, b))
    assert a.randint(1 << 30) == b.rs.randint(1 << 30)
