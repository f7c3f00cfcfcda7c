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
