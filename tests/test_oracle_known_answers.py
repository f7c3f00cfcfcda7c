"""Hand-derived known answers for the restated torch_geometric 2.3.0 leaf ops
(oracle/pyg_ops.py).  These are the only pin those ops have (PyG is absent)."""
import math

import torch

from oracle import pyg_ops as P
from oracle import path as O


def test_sage_mean_integer_graph():
    # 4 nodes on a path 0-1-2-3 (both directions), node features = [i, 10*i]; identity weights.
    x = torch.tensor([[0., 0.], [1., 10.], [2., 20.], [3., 30.]])
    ei = torch.tensor([[1, 0, 2, 1, 3, 2], [0, 1, 1, 2, 2, 3]])
    I = torch.eye(2)
    out = P.sage_conv(x, ei, I, torch.zeros(2), 2 * I, None, None, "mean")
    # agg = mean of neighbours: [1,10], [1,10], [2,20], [2,20]; out = agg + 2*x
    exp = torch.tensor([[1., 10.], [3., 30.], [6., 60.], [8., 80.]])
    assert torch.equal(out, exp)


def test_sage_mean_isolated_node_gets_zero_and_project_uses_relu():
    x = torch.tensor([[1., -2.], [3., 4.], [-5., 6.]])
    ei = torch.tensor([[0], [1]])  # only 0 -> 1
    I = torch.eye(2)
    out = P.sage_conv(x, ei, I, torch.tensor([0.5, 0.5]), torch.zeros(2, 2), I, torch.zeros(2), "mean")
    # projected source relu(x0) = [1, 0] arrives at node 1; nodes 0 and 2 have no in-edge -> agg 0
    exp = torch.tensor([[0.5, 0.5], [1.5, 0.5], [0.5, 0.5]])
    assert torch.equal(out, exp)


def test_sage_max_with_self_loops_and_lin_r_on_unprojected_x():
    x = torch.tensor([[1., 9.], [5., 2.], [3., 3.]])
    ei = P.add_remaining_self_loops(torch.tensor([[0, 1, 2], [2, 2, 2]]), 3)  # existing loop 2->2 is replaced
    assert ei.tolist() == [[0, 1, 0, 1, 2], [2, 2, 0, 1, 2]]
    I = torch.eye(2)
    out = P.sage_conv(x, ei, I, None, 10 * I, None, None, "max")
    exp = torch.tensor([[1. + 10, 9. + 90], [5. + 50, 2. + 20], [5. + 30, 9. + 30]])
    assert torch.equal(out, exp)


def test_graph_layer_norm_uses_global_stats_and_eps_on_std():
    x = torch.tensor([[1., 2.], [3., 6.]])
    mean = 3.0
    std = math.sqrt(((1 - 3) ** 2 + (2 - 3) ** 2 + 0 + (6 - 3) ** 2) / 4)
    w, b = torch.tensor([2., 1.]), torch.tensor([0.5, -0.5])
    exp = (x - mean) / (std + 1e-5) * w + b
    torch.testing.assert_close(P.graph_layer_norm(x, w, b), exp, rtol=1e-6, atol=1e-6)
    # per-row LayerNorm would give +-1 in every row: make sure we are NOT that
    assert not torch.allclose(P.graph_layer_norm(x, torch.ones(2), torch.zeros(2)),
                              torch.nn.functional.layer_norm(x, (2,)))


def test_positional_encoding_frequencies():
    f = P.positional_encoding_frequency(8)
    exp = torch.tensor([1e-4 ** (i / 3) for i in range(4)])
    torch.testing.assert_close(f, exp, rtol=1e-6, atol=0)
    pe = P.positional_encoding(torch.tensor([0, 2]), f)
    assert pe.shape == (2, 8)
    assert torch.equal(pe[0], torch.tensor([0., 0., 0., 0., 1., 1., 1., 1.]))
    torch.testing.assert_close(pe[1, 0], torch.sin(torch.tensor(2.0)))
    torch.testing.assert_close(pe[1, 4], torch.cos(torch.tensor(2.0)))


def test_radius_band_edges():
    ei = O.temporal_radius_edges(torch.arange(5) - 2, k=1)
    pairs = set(map(tuple, ei.t().tolist()))
    assert pairs == {(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2), (3, 4), (4, 3)}
    ei2 = O.temporal_radius_edges(torch.arange(4), k=2)
    assert ei2.shape[1] == 2 * (3 + 2)  # E = 2kN - k(k+1) = 16 - 6 = 10
    # grouped by target
    assert ei[1].tolist() == sorted(ei[1].tolist())


def test_radius_respects_batch_and_collate_offsets():
    a = P.OData(x=torch.zeros(3, 1), pos=torch.arange(3), y=1, edge_index=O.temporal_radius_edges(torch.arange(3), 1))
    b = P.OData(x=torch.zeros(2, 1), pos=torch.arange(2), y=0, edge_index=O.temporal_radius_edges(torch.arange(2), 1))
    batch = P.collate([a, b])
    assert batch.batch.tolist() == [0, 0, 0, 1, 1]
    assert batch.y.tolist() == [1, 0]
    assert batch.ptr.tolist() == [0, 3, 5]
    assert set(map(tuple, batch.edge_index.t().tolist())) == {(0, 1), (1, 0), (1, 2), (2, 1), (3, 4), (4, 3)}


def test_lta_connectivity_known_answer_and_verb_zero_quirk():
    pos = torch.arange(5)
    y = torch.tensor([[-1, -1], [-1, -1], [3, 1], [0, 2], [4, 1]])  # verb 0 at node 3 is NOT counted
    ei = O.lta_temporal_connectivity(pos, y, 1.5)
    pairs = set(map(tuple, ei.t().tolist()))
    band = {(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2), (3, 4), (4, 3)}
    # n_in=2, n_f=2 (not 3): last floor(1.5)=1 input clip (node 1) -> forecast nodes 2,3 only
    assert pairs == band | {(1, 2), (1, 3)}
    keys = (ei[0] * 5 + ei[1]).tolist()
    assert keys == sorted(set(keys))  # coalesced


def test_global_max_pool_and_scatter_sum():
    x = torch.tensor([[1., -5.], [3., -7.], [-2., -1.]])
    out = P.global_max_pool(x, torch.tensor([0, 0, 1]))
    assert torch.equal(out, torch.tensor([[3., -5.], [-2., -1.]]))
    s = P.scatter_sum(x, torch.tensor([2, 2, 0]), 4)
    assert torch.equal(s, torch.tensor([[-2., -1.], [0., 0.], [4., -12.], [0., 0.]]))


def test_ce_ignore_index_rows_are_zero_but_count_in_mean():
    logits = (torch.zeros(4, 2), torch.zeros(4, 4))
    y = torch.tensor([[-1, -1], [0, 1], [-1, -1], [-1, -1]])
    loss = O.multihead_ce(logits, y)
    exp = math.log(2) + math.log(4)
    torch.testing.assert_close(loss, torch.tensor([0., exp, 0., 0.]))
    torch.testing.assert_close(loss.mean(), torch.tensor(exp / 4))


def test_compute_edges_known_answer():
    bank = torch.tensor([[1., 0.], [0., 1.], [-1., 0.], [1., 1.]])
    f = torch.tensor([[2., 0.1], [0.1, 3.]])
    edges, closest = O.compute_edges(f, bank, k=2)
    assert closest.tolist() == [[0, 3], [1, 3]]
    assert edges.tolist() == [[0, 3, 1, 3], [4, 4, 5, 5]]
