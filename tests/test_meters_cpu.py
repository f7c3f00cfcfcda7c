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
