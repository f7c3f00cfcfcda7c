#!/usr/bin/env python3
"""tests/golden/trn_multiscale.pt: the reference's ``RelationModuleMultiScale`` (models/TRN.py:9-74) run as it is -- the class
imports with no stand-ins -- on seeded parameters and inputs: structure (scales, relation sets, sub-sampling), state dict,
forward output and the gradients of a seeded cotangent.  Build container only.  TEST INFRASTRUCTURE ONLY.
Usage: python oracle/make_golden_trn_multiscale.py"""
import importlib.util
from math import ceil
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
spec = importlib.util.spec_from_file_location("_ref_TRN", "/root/reference/models/TRN.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

cases = []
for seed, (D, NB, F, B) in enumerate([(48, 32, 5, 7), (64, 64, 3, 9), (24, 16, 8, 4), (32, 40, 2, 5)]):
    torch.manual_seed(100 + seed)
    m = ref.RelationModuleMultiScale(D, NB, F)
    x = torch.randn(B, F, D, requires_grad=True)
    cot = torch.randn(B, len(m.scales), NB)
    out = m(x)
    (out * cot).sum().backward()
    selected = []
    for sid in range(len(m.scales)):
        if sid == 0:
            selected.append([list(m.relations_scales[0][0])])
        else:
            n_tot, n_sel = len(m.relations_scales[sid]), m.subsample_scales[sid]
            selected.append([list(m.relations_scales[sid][int(ceil(i * n_tot / n_sel))]) for i in range(n_sel)])
    cases.append({"seed": 100 + seed, "img_feature_dim": D, "num_bottleneck": NB, "num_frames": F,
                  "scales": list(m.scales), "subsample_scales": list(m.subsample_scales),
                  "relations_scales": [[list(r) for r in rs] for rs in m.relations_scales], "selected": selected,
                  "state_dict": {k: v.clone() for k, v in m.state_dict().items()},
                  "x": x.detach().clone(), "cot": cot, "out": out.detach().clone(), "dx": x.grad.clone(),
                  "grads": {k: p.grad.clone() for k, p in m.named_parameters()}})
torch.save({"cases": cases}, REPO / "tests" / "golden" / "trn_multiscale.pt")
print("trn_multiscale.pt:", [(c["num_frames"], c["scales"], c["subsample_scales"]) for c in cases])
