"""Oracle (oracle/path.py) vs the golden vectors produced by the REFERENCE's own Python
(oracle/make_golden.py).  CPU only.  Tolerances: fp32, same op order => 1e-5 abs / 1e-5 rel
(the oracle uses the same torch kernels, so most comparisons are in fact exact)."""
import torch
import pytest

from oracle import path as O
from oracle import pyg_ops as P

TOL = dict(rtol=1e-5, atol=1e-5)


def trainable(key):
    """Buffers / frozen tensors of the reference layout: the PE frequency buffer
    (models/graph.py:37) and the frozen prototype embeddings (graphONE.py:47-49)."""
    return not (key.endswith("positional_encoding.frequency") or key.startswith("embeddings."))


def leafify(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.clone().requires_grad_(True) if (v.is_floating_point() and trainable(k)) else v.clone()
    return out


def data_of(d):
    return P.OData(**d)


def test_trn_pooling(golden):
    G = golden("trn_pooling")
    sd = leafify(G["sd"])
    x = G["x"].clone().requires_grad_(True)
    out = O.trn_pooling(sd, x)
    torch.testing.assert_close(out, G["out"], **TOL)
    (out * G["w"]).sum().backward()
    torch.testing.assert_close(x.grad, G["grad_x"], **TOL)
    for k, g in G["grads"].items():
        torch.testing.assert_close(sd[k].grad, g, **TOL)


@pytest.mark.parametrize("case", ["ar_T9_k1", "lta_T22_k1", "oscc_T4_k2", "pnr_T16_k2"])
def test_graph_forward(golden, case):
    G = golden("graph_forward")
    c = G["cases"][case]
    sd = leafify(G["sd"])
    d = c["data"]
    out = O.graph_forward(sd, d["x"], d["pos"], d["edge_index"], G["depth"])
    torch.testing.assert_close(out, c["out"], **TOL)
    (out * c["w"]).sum().backward()
    for k, g in c["grads"].items():
        torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("key", ["ar_avg0", "ar_avg1", "lta_avg0", "lta_avg1"])
def test_multihead_tasks(golden, key):
    G = golden("heads")
    c = G[key]
    f = O.projection_features(c["sd"], G["feat"])
    torch.testing.assert_close(f, c["features"], **TOL)
    plain = O.multihead_logits(c["sd"], f, 2)
    fused = O.multihead_logits(c["sd"], f, 2, c["aux"], average_logits=key.endswith("1"))
    for a, b in zip(plain, c["logits"]):
        torch.testing.assert_close(a, b, **TOL)
    for a, b in zip(fused, c["logits_fused"]):
        torch.testing.assert_close(a, b, **TOL)
    torch.testing.assert_close(O.multihead_ce(fused, G["y2"]), c["loss"], **TOL)


def test_metric_selector(golden):
    G = golden("heads")
    torch.testing.assert_close(O.multihead_ce(G["selector"]["logits"], G["y2"]), G["selector"]["loss"], **TOL)
    with pytest.raises(ValueError):
        O.multihead_ce(G["selector"]["logits"][:1], G["y2"])


@pytest.mark.parametrize("kind", ["ce", "bce"])
@pytest.mark.parametrize("avg", [0, 1])
def test_oscc(golden, kind, avg):
    G = golden("heads")
    c = G[f"oscc_{kind}_avg{avg}"]
    f = O.projection_features(c["sd"], G["feat"])
    torch.testing.assert_close(O.oscc_logits(c["sd"], f, G["batch"]), c["logits"], **TOL)
    fused = O.oscc_logits(c["sd"], f, G["batch"], c["aux"], bool(avg))
    torch.testing.assert_close(fused, c["logits_fused"], **TOL)
    torch.testing.assert_close(O.oscc_loss(fused, c["y"], kind), c["loss"], **TOL)


@pytest.mark.parametrize("avg", [0, 1])
def test_pnr(golden, avg):
    G = golden("heads")
    c = G[f"pnr_avg{avg}"]
    f = O.projection_features(c["sd"], G["feat"])
    torch.testing.assert_close(O.pnr_logits(c["sd"], f), c["logits"], **TOL)
    fused = O.pnr_logits(c["sd"], f, c["aux"], bool(avg))
    torch.testing.assert_close(fused, c["logits_fused"], **TOL)
    torch.testing.assert_close(O.pnr_loss(fused, c["y"]), c["loss"], **TOL)


@pytest.mark.parametrize("residual", [0, 1])
def test_graphone(golden, residual):
    G = golden("graphone")
    c = G[f"residual{residual}"]
    sd = leafify(c["sd"])
    feats = {t: f.clone().requires_grad_(True) for t, f in c["features"].items()}
    out, closest = O.graphone_interact(sd, feats, G["k"], c["depth"], bool(residual))
    for t in feats:
        torch.testing.assert_close(out[t], c["out"][t], **TOL)
        for a, b in zip(closest[t], c["closest"][t]):
            assert torch.equal(a, b)  # index op: bit-exact
    sum((out[t] * c["w"][t]).sum() for t in feats).backward()
    for t in feats:
        torch.testing.assert_close(feats[t].grad, c["grad_features"][t], **TOL)
    for k, g in c["grads"].items():
        torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-5)


def test_cos_dissimilarity_shape(golden):
    G = golden("graphone")
    assert G["cos_example"].shape == (5, G["banks"]["ar"].shape[0])


def test_build_graphone(golden):
    G = golden("build_graphone")
    banks = O.build_graphone(G["backbone"], G["tasks"], [data_of(b) for b in G["batches"]], G["n_classes"])
    assert set(banks) == set(G["banks"])
    for t in banks:
        assert banks[t].dtype == torch.float32
        torch.testing.assert_close(banks[t], G["banks"][t], **TOL)


@pytest.mark.parametrize("name", ["lta_T22", "lta_T22_verb0", "lta_T8_verb0_first", "lta_T12_r2.5"])
def test_lta_connectivity_exact(golden, name):
    c = golden("edges_loader")[name]
    ei = O.lta_temporal_connectivity(c["pos"], c["y"], c["r"])
    assert torch.equal(ei, c["edge_index"])  # integer op: bit-exact incl. coalesced order


def _adam_steps(G, objective, n_steps, groups):
    params = {g: leafify(G["before"][g]) for g in groups}
    flat = [p for g in groups for p in params[g].values() if p.requires_grad]
    opt = torch.optim.Adam(flat, lr=G["lr"], weight_decay=G["weight_decay"])
    losses = []
    for it in range(n_steps):
        opt.zero_grad()
        total, detail = objective(params, it)
        total.backward()
        opt.step()
        losses.append(detail)
    return params, losses


def test_mtl_train_two_iterations(golden):
    """Oracle objective + torch.optim.Adam reproduces the reference main_temporal.train loop."""
    G = golden("mtl_train")
    groups = ["temporal_graph", "task/recognition", "task/oscc", "task/lta", "task/pnr"]
    names = {"ar": "task/recognition", "oscc": "task/oscc", "lta": "task/lta", "pnr": "task/pnr"}

    def objective(params, it):
        batches = {t: data_of(G["batches"][t][it]) for t in ("ar", "lta", "oscc", "pnr")}
        return O.mtl_objective(params["temporal_graph"], {t: params[n] for t, n in names.items()},
                               batches, G["weights"])

    params, losses = _adam_steps(G, objective, 2, groups)
    for it in range(2):
        for t in ("ar", "lta", "pnr"):
            torch.testing.assert_close(losses[it][t][1], G["loss_vectors"][t][it], rtol=1e-4, atol=1e-5)
    assert G["loss_vectors"]["oscc"] == []  # weight 0 => loader skipped by multiloader
    for g in groups:
        for k, v in G["after"][g].items():
            torch.testing.assert_close(params[g][k].detach(), v, rtol=1e-4, atol=2e-6, msg=lambda m: f"{g}/{k}: {m}")
    # the disabled task never received a gradient: parameters untouched (Adam skips grad=None)
    for k, v in G["before"]["task/oscc"].items():
        assert torch.equal(G["after"]["task/oscc"][k], v)


def test_egopack_train_two_iterations(golden):
    G = golden("egopack_train")
    groups = ["temporal_graph", "task/recognition", "task/oscc", "task/lta", "task/pnr", "graphone"]
    names = {"ar": "task/recognition", "oscc": "task/oscc", "lta": "task/lta", "pnr": "task/pnr"}

    def objective(params, it):
        d = data_of(G["batches"]["oscc"][it])
        feat = O.graph_forward(params["temporal_graph"], d.x, d.pos, d.edge_index, 3)
        loss, logits, aux, closest = O.egopack_task_loss(
            "oscc", {t: params[n] for t, n in names.items()}, params["graphone"], feat, d.batch, d.y,
            ["ar", "lta", "pnr"], G["k"], G["depth"], G["residual"], average_logits=True, num_graphs=d.num_graphs)
        return 1.0 * loss.mean(), loss

    params, losses = _adam_steps(G, objective, 2, groups)
    for it in range(2):
        torch.testing.assert_close(losses[it], G["loss_vectors"]["oscc"][it], rtol=1e-4, atol=1e-5)
    for g in groups:
        for k, v in G["after"][g].items():
            torch.testing.assert_close(params[g][k].detach(), v, rtol=1e-4, atol=2e-6, msg=lambda m: f"{g}/{k}: {m}")
    # frozen banks unchanged
    for t in ("ar", "lta", "pnr"):
        assert torch.equal(G["after"]["graphone"][f"embeddings.{t}.weight"], G["before"]["graphone"][f"embeddings.{t}.weight"])


# ---- variants of the cited functions beside the configured ones (tests/golden/variants.pt) ---------------------------
@pytest.mark.parametrize("name", ["l2", "trainable", "l2_trainable"])
def test_graphone_variants(golden, name):
    """distance_func='l2' (graphONE.py:126-127) and trainable prototypes (freeze=False, :47-49): outputs, exact
    nearest-prototype indices, feature / stage gradients and -- when trainable -- the prototype gradients."""
    G = golden("variants")
    c = G[name]
    free = not c["kw"]["freeze"]
    sd = {k: (v.clone().requires_grad_(True) if (not k.startswith("embeddings.") or free) else v.clone()) for k, v in c["sd"].items()}
    feats = {t: f.clone().requires_grad_(True) for t, f in c["features"].items()}
    out, closest = O.graphone_interact(sd, feats, G["k"], G["depth"], True, c["kw"]["distance_func"])
    for t in feats:
        torch.testing.assert_close(out[t], c["out"][t], **TOL)
        for a, b in zip(closest[t], c["closest"][t]):
            assert torch.equal(a, b)
    sum((out[t] * c["w"][t]).sum() for t in feats).backward()
    for t in feats:
        torch.testing.assert_close(feats[t].grad, c["grad_features"][t], **TOL)
    assert any(k.startswith("embeddings.") for k in c["grads"]) == free
    for k, g in c["grads"].items():
        torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-5)


def test_oscc_bce_gradients(golden):
    c = golden("variants")["oscc_bce"]
    sd = leafify(c["sd"])
    logits = O.oscc_logits(sd, O.projection_features(sd, c["feat"]), c["batch"])
    loss = O.oscc_loss(logits, c["y"], "bce")
    torch.testing.assert_close(loss, c["loss"], **TOL)
    (loss * c["w"]).sum().backward()
    for k, g in c["grads"].items():
        torch.testing.assert_close(sd[k].grad, g, rtol=1e-4, atol=1e-5)


def test_focal_loss_known_answers():
    """torchvision.ops.sigmoid_focal_loss(alpha=0.5, gamma=2) (absent package: restated, parity unpinned) on hand-
    computable points: x = 0 -> p = 1/2: 0.5 * (1/2)^2 * ln 2 for either class; a confident correct logit gives ~0, a
    confident wrong one ~alpha * |x|."""
    import math
    y = torch.tensor([1, 0])
    l = O.oscc_loss(torch.zeros(2, 2), y, "focal")
    assert torch.allclose(l, torch.full((2, 2), 0.5 * 0.25 * math.log(2.0)), atol=1e-7)
    l = O.oscc_loss(torch.tensor([[-20.0, 20.0], [-20.0, 20.0]]), y, "focal")  # row 0 right on both columns, row 1 wrong
    assert l[0].abs().max() < 1e-12 and torch.allclose(l[1], torch.full((2,), 0.5 * 20.0), rtol=1e-6)
