#!/usr/bin/env python3
"""Where does the HIP path's bf16-mode BACKWARD leave the oracle's storage model (oracle/storage.py)?  Stage by stage:
TRN pooling alone, the whole backbone, backbone + projection head, each driven by a random f32 cotangent, parameter gradients
compared tensor by tensor against the oracle with and without the storage model.  (development tool, GPU box)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from egopack_amd import data as D, ops  # noqa: E402
from egopack_amd.models import Graph  # noqa: E402
from egopack_amd.models.tasks import RecognitionTask  # noqa: E402
from oracle import path as O, pyg_ops as P, storage as S  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))


def main():
    torch.manual_seed(0)
    F_IN, Sg, H, B, T = 128, 3, 256, 16, 32
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
    model = Graph(F_IN, hidden_size=H, depth=3, temporal_pooling=trn, num_segments=Sg)
    task = RecognitionTask(H, H, (13, 17))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    tsd = {k: v.clone() for k, v in task.state_dict().items()}
    ds = D.SyntheticTaskDataset("ar", B, T, Sg, F_IN, (13, 17), k=1, seed=5)
    host = D.collate([ds[i] for i in range(B)])
    host.x = host.x.to(torch.bfloat16)
    od = P.OData(x=host.x.float(), pos=host.pos, edge_index=host.edge_index, batch=host.batch, y=host.y, num_graphs=B)
    N = host.x.shape[0]
    R = torch.randn(N, H)
    model.cuda().train()
    task.cuda().train()
    dev = host.to("cuda")
    stages = {
        "trn": (lambda: model.temporal_pooling(dev.x, None, dev.pos),
                lambda l: O.trn_pooling(O._sub(l["m"], "temporal_pooling."), od.x)),
        "backbone": (lambda: model(dev), lambda l: O.graph_forward(l["m"], od.x, od.pos, od.edge_index, 3)),
        "backbone+proj": (lambda: task.forward_features(model(dev)),
                          lambda l: O.projection_features(l["t"], O.graph_forward(l["m"], od.x, od.pos, od.edge_index, 3))),
    }
    for name, (hip, orc) in stages.items():
        with ops.compute_mode("bf16"):
            for p in [*model.parameters(), *task.parameters()]:
                p.grad = None
            out = hip()
            out.backward(R.cuda().to(out.dtype))
            torch.cuda.synchronize()
        got = {"m." + k: p.grad.float().cpu() for k, p in model.named_parameters() if p.grad is not None}
        got.update({"t." + k: p.grad.float().cpu() for k, p in task.named_parameters() if p.grad is not None})
        res = {}
        for storage in (False, True):
            leaf = {"m": {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("frequency") else v) for k, v in sd.items()},
                    "t": {k: v.clone().requires_grad_(True) for k, v in tsd.items()}}
            with S.bf16_storage(storage):
                o = orc(leaf)
                (o * (R.to(torch.bfloat16).float() if storage else R)).sum().backward()
            res[storage] = ({f"{g}.{k}": v.grad for g, d in leaf.items() for k, v in d.items() if v.requires_grad and v.grad is not None},
                            o.detach())
        print(f"== {name}: forward vs f32 {rel(out.float().cpu(), res[False][1]):.2e}, vs model {rel(out.float().cpu(), res[True][1]):.2e}")
        for k in got:
            if k in res[True][0]:
                print(f"   {k:45s} vs f32 {rel(got[k], res[False][0][k]):.4f}   vs model {rel(got[k], res[True][0][k]):.4f}")


if __name__ == "__main__":
    main()
