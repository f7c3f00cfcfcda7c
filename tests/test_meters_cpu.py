"""Validation meters, CPU side: the oracle restatement against the fixtures produced by the reference's own
meter code (tests/golden/meters.pt, oracle/make_golden_meters.py) and against hand-computable known answers for
the leaves that live in absent third-party packages (torchmetrics, editdistance: parity unpinned, see oracle/meters.py)."""
import numpy as np
import pytest
import torch

from oracle import meters as OM


def test_topk_accuracy_and_recall_match_reference_functions(golden):
    g = golden("meters")["topk"]
    scores, labels = g["scores"].numpy(), g["labels"].numpy()
    assert OM.topk_accuracy(scores, labels, g["ks"]) == pytest.approx(g["accuracy"], abs=1e-12)
    for k, v in g["recall"].items():
        assert OM.topk_recall(scores, labels, k) == pytest.approx(v, abs=1e-12)
    assert OM.topk_accuracy(scores, labels, (1, 5), selected_class=3) == pytest.approx(g["class3"], abs=1e-12)
    # the rank formulation used on the device gives the same counts
    rank = OM.label_rank(scores, labels)
    for k, acc in zip(g["ks"], g["accuracy"]):
        assert float((rank < k).mean()) == pytest.approx(acc, abs=1e-12)
        assert OM.multiclass_accuracy(scores, labels, k) == pytest.approx(acc, abs=1e-12)
    assert OM.multiclass_accuracy(scores, labels, 5, average="macro") == pytest.approx(g["recall"][5], abs=1e-12)


def test_pnr_localisation_error_matches_reference_meter(golden):
    g = golden("meters")["pnr"]
    ptr = np.concatenate([[0], np.cumsum(np.bincount(g["batch"].numpy()))])
    probs = torch.sigmoid(g["logits"]).numpy()
    err = OM.pnr_localisation_errors(probs, ptr, g["start_frame"].numpy(), g["end_frame"].numpy(), g["pnr_frame"].numpy())
    np.testing.assert_allclose(err, g["loc_errors"].numpy(), rtol=0, atol=1e-12)


def test_lta_edit_distance_bookkeeping_matches_reference_meter(golden):
    g = golden("meters")["lta"]
    for head, key in ((0, "verbs"), (1, "nouns")):
        d = OM.lta_edit_distance(g["predictions"][head].numpy(), g["labels"][:, head].numpy())
        np.testing.assert_allclose(d, g[key].numpy(), rtol=0, atol=1e-12)
    assert g["verbs"][0] == 0  # the planted exact sample


def test_levenshtein_known_answers():
    assert OM.levenshtein("kitten", "sitting") == 3
    assert OM.levenshtein("flaw", "lawn") == 2
    assert OM.levenshtein([1, 2, 3], [1, 2, 3]) == 0
    assert OM.levenshtein([], [4, 5]) == 2
    assert OM.levenshtein([1, 2], [2, 1]) == 2  # a transposition costs 2: plain Levenshtein, not Damerau


def test_multiclass_accuracy_known_answers():
    scores = np.array([[0.1, 0.9, 0.0], [0.8, 0.1, 0.1], [0.2, 0.3, 0.5], [0.5, 0.4, 0.1]])
    labels = np.array([1, 1, -1, 0])
    assert OM.multiclass_accuracy(scores, labels, 1) == pytest.approx(2 / 3)  # ignored row leaves the denominator
    assert OM.multiclass_accuracy(scores, labels, 2) == pytest.approx(1.0)  # row 1: label 1 ties with class 2, the lower index ranks first
    labels2 = np.array([1, 2, -1, 0])
    assert OM.multiclass_accuracy(scores, labels2, 2) == pytest.approx(2 / 3)  # ... so class 2 of row 1 is rank 2: a miss at k = 2
    np.testing.assert_allclose(OM.multiclass_accuracy(scores, labels, 1, average=None), [1.0, 0.5, 0.0])
    assert OM.multiclass_accuracy(scores, labels, 1, average="macro") == pytest.approx(0.75)  # class 2 has no support


def test_binary_stats_and_auroc_known_answers():
    p = np.array([0.9, 0.8, 0.5, 0.4, 0.3, 0.3])
    t = np.array([1, 0, 1, 1, 0, 0])
    s = OM.binary_stats(p, t)
    assert s["accuracy"] == pytest.approx(3 / 6) and s["recall"] == pytest.approx(1 / 3)  # 0.5 is NOT > 0.5
    # pairs (pos, neg): 0.9>{.8,.3,.3}=3, 0.5>{.3,.3}=2, 0.4>{.3,.3}=2 -> 7/9
    assert OM.binary_auroc(p, t) == pytest.approx(7 / 9)
    assert OM.binary_auroc(np.array([0.5, 0.5]), np.array([1, 0])) == pytest.approx(0.5)


def test_calibration_error_known_answers_and_the_device_sums():
    """MulticlassCalibrationError as the reference builds it (utils/meters/ego4d.py:52-53,66-67; torchmetrics 1.0.1 is absent:
    parity unpinned, so hand-derived answers): probabilities are taken as they are, logits go through a softmax, ignored rows
    drop out, the 15-bin l1 error weighs |accuracy - confidence| by the bin shares, one bin / l2 is |mean acc - mean conf|."""
    p = np.array([[0.9, 0.1], [0.6, 0.4], [0.3, 0.7], [0.2, 0.8], [0.5, 0.5]], dtype=np.float32)
    y = np.array([0, 1, 1, 0, -1])
    # bins of width 1/15: conf 0.9 -> bin 13 (hit), 0.6 -> bin 9 (miss), 0.7 -> bin 10 (hit), 0.8 -> bin 12 (miss); row 4 ignored
    want = (abs(1 - 0.9) + abs(0 - 0.6) + abs(1 - 0.7) + abs(0 - 0.8)) / 4
    assert OM.multiclass_calibration_error(p, y) == pytest.approx(want, abs=1e-6)
    assert OM.multiclass_calibration_error(p, y, n_bins=1, norm="l2") == pytest.approx(abs(0.5 - 0.75), abs=1e-6)
    assert OM.multiclass_calibration_error(p, y, norm="max") == pytest.approx(0.8, abs=1e-6)
    sure = np.array([[1.0, 0.0], [1.0, 0.0]], dtype=np.float32)  # confidence exactly 1: a bin of its own, perfectly calibrated
    assert OM.multiclass_calibration_error(sure, np.array([0, 0])) == 0.0
    logits = np.log(np.array([[0.9, 0.1], [0.6, 0.4], [0.3, 0.7], [0.2, 0.8]], dtype=np.float64)) + 3.0  # softmax gives p back
    assert OM.multiclass_calibration_error(logits, y[:4]) == pytest.approx(want, abs=1e-6)
    assert OM.multiclass_calibration_error(p[:0], y[:0]) == 0.0
    # the product meter's per-bin sums (plain tensor ops: runs on the CPU too), fed in two pieces, equal the one-shot oracle
    from egopack_amd.meters import _Calibration
    gen = torch.Generator().manual_seed(11)
    s = torch.randn(500, 13, generator=gen) * 2
    t = torch.randint(-1, 13, (500,), generator=gen)
    for n_bins, norm in ((15, "l1"), (1, "l2"), (15, "max")):
        c = _Calibration(n_bins, norm, "cpu")
        c.update(s[:200], t[:200])
        c.update(s[200:], t[200:])
        assert c.compute() == pytest.approx(OM.multiclass_calibration_error(s.numpy(), t.numpy(), n_bins, norm), abs=1e-6)
    probs = torch.rand(40, 13, generator=gen) * 0.9  # scores inside [0, 1]: taken as they are, no softmax
    other = _Calibration(15, "l1", "cpu")
    other.update(probs, t[:40])
    assert other.compute() == pytest.approx(OM.multiclass_calibration_error(probs.numpy(), t[:40].numpy()), abs=1e-6)
