"""Host half of the resident input pipeline: the segment-sampling index arithmetic against the
reference's own functions run on seeded numpy state (tests/golden/sampling.pt, oracle/make_golden_sampling.py)."""
import numpy as np
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
