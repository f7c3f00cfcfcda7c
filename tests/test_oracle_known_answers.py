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


# ---- independent second formulations: a slip in a restated leaf cannot pass both its own statement and these ----
def _dense_adjacency(edge_index, n):
    a = torch.zeros(n, n, dtype=torch.float64)
    a.index_put_((edge_index[1], edge_index[0]), torch.ones(edge_index.shape[1], dtype=torch.float64), accumulate=True)
    return a  # a[i, j] = number of edges j -> i


def test_sage_mean_equals_the_dense_normalised_adjacency_product():
    """SAGEConv(mean, root_weight, project) as D^-1 A relu(X Wp^T + bp) Wl^T + bl + X Wr^T with a dense adjacency (multi-edges
    counted, isolated nodes: zero row) -- no scatter, no index_select."""
    gen = torch.Generator().manual_seed(21)
    n, c, h = 23, 6, 5
    x = torch.randn(n, c, generator=gen, dtype=torch.float64)
    ei = torch.randint(0, n, (2, 70), generator=gen)
    ei = ei[:, ei[1] != 7]  # node 7 has no in-edge
    wl, bl, wr = torch.randn(h, c, generator=gen, dtype=torch.float64), torch.randn(h, generator=gen, dtype=torch.float64), torch.randn(h, c, generator=gen, dtype=torch.float64)
    wp, bp = torch.randn(c, c, generator=gen, dtype=torch.float64), torch.randn(c, generator=gen, dtype=torch.float64)
    a = _dense_adjacency(ei, n)
    norm = a / a.sum(1, keepdim=True).clamp(min=1)
    want = norm @ torch.relu(x @ wp.t() + bp) @ wl.t() + bl + x @ wr.t()
    torch.testing.assert_close(P.sage_conv(x, ei, wl, bl, wr, wp, bp, "mean"), want, rtol=1e-12, atol=1e-12)
    want_plain = norm @ x @ wl.t() + x @ wr.t()
    torch.testing.assert_close(P.sage_conv(x, ei, wl, None, wr, None, None, "mean"), want_plain, rtol=1e-12, atol=1e-12)
    assert bool((norm[7] == 0).all())


def test_sage_max_equals_a_masked_dense_maximum():
    gen = torch.Generator().manual_seed(22)
    n, c = 17, 4
    x = torch.randn(n, c, generator=gen, dtype=torch.float64)
    ei = P.add_remaining_self_loops(torch.randint(0, n, (2, 40), generator=gen), n)
    wl, wr = torch.randn(3, c, generator=gen, dtype=torch.float64), torch.randn(3, c, generator=gen, dtype=torch.float64)
    mask = _dense_adjacency(ei, n) > 0  # [target, source]
    m = torch.where(mask[:, :, None], x[None, :, :], torch.full((), -float("inf"), dtype=torch.float64)).max(1).values
    torch.testing.assert_close(P.sage_conv(x, ei, wl, None, wr, None, None, "max"), m @ wl.t() + x @ wr.t(), rtol=1e-12, atol=1e-12)


def test_graph_layer_norm_equals_layer_norm_over_the_flattened_tensor_with_the_eps_moved_to_the_std():
    """mode='graph' = ONE normalisation over all N * C elements.  F.layer_norm over the flattened tensor is that with eps inside
    the square root: y_ln = (x - m) / sqrt(v + e); PyG divides by (sqrt(v) + e) -- so y = y_ln * sqrt(v + e) / (sqrt(v) + e)."""
    gen = torch.Generator().manual_seed(23)
    x = torch.randn(19, 8, generator=gen, dtype=torch.float64) * 3 + 1.5
    w, b = torch.randn(8, generator=gen, dtype=torch.float64), torch.randn(8, generator=gen, dtype=torch.float64)
    eps = 1e-5
    y_ln = torch.nn.functional.layer_norm(x.reshape(-1), (x.numel(),), eps=eps).reshape(x.shape)
    v = x.var(unbiased=False)
    want = y_ln * torch.sqrt(v + eps) / (torch.sqrt(v) + eps) * w + b
    torch.testing.assert_close(P.graph_layer_norm(x, w, b, eps), want, rtol=1e-12, atol=1e-12)
    # the eps placement is visible on a nearly constant tensor: the two conventions differ by orders of magnitude there
    tiny = torch.full((4, 8), 2.0, dtype=torch.float64) + 1e-7 * torch.randn(4, 8, generator=gen, dtype=torch.float64)
    got = P.graph_layer_norm(tiny, torch.ones(8, dtype=torch.float64), torch.zeros(8, dtype=torch.float64), eps)
    inside = torch.nn.functional.layer_norm(tiny.reshape(-1), (32,), eps=eps).reshape(4, 8)
    assert float(got.abs().max()) > 100 * float(inside.abs().max())  # (sqrt(v + e) / (sqrt(v) + e) ~ 300 at std 1e-7)


def test_positional_encoding_equals_the_complex_exponential():
    f = P.positional_encoding_frequency(16)
    assert float(f[0]) == 1.0 and abs(float(f[-1]) - 1e-4) < 1e-10  # logspace(0, 1, C/2, base): base^0 .. base^1
    pos = torch.tensor([-3.0, 0.0, 1.0, 7.5])
    z = torch.exp(1j * (pos.double()[:, None] * f.double()[None, :]))
    want = torch.cat([z.imag, z.real], -1).float()  # [sin | cos]
    torch.testing.assert_close(P.positional_encoding(pos, f), want, rtol=1e-6, atol=1e-6)


def test_radius_graph_equals_a_brute_force_strict_test():
    """torch_cluster's radius test is strict: at the reference's r = k + 0.5 on integer positions both conventions
    agree; at an integer r they differ -- the restatement must be the strict one."""
    pos = torch.tensor([0, 1, 2, 4, 0, 1])
    batch = torch.tensor([0, 0, 0, 0, 1, 1])
    for r in (1.5, 2.0, 2.5):
        want = {(j, i) for i in range(6) for j in range(6) if i != j and int(batch[i]) == int(batch[j]) and abs(int(pos[i]) - int(pos[j])) < r}
        got = set(map(tuple, P.radius_graph(pos, r, batch).t().tolist()))
        assert got == want, r
    assert (2, 0) not in set(map(tuple, P.radius_graph(pos, 2.0, batch).t().tolist()))  # |0 - 2| = r: not an edge
