#!/usr/bin/env python3
"""TWO ranks really stepping together, on one GPU, against ONE process on the global batch.

The pool's boxes have one GPU and RCCL refuses two ranks on one device, so the two rank processes share ``cuda:0`` and
exchange through a gloo group (dist.all_reduce_sum_ bounces device tensors through host memory on gloo groups): everything
of the data-parallel step is the product path -- batch sharding, the HIP kernels, engine.MTLStep with a GradSync of world
size 2, FlatAdam's 1 / world gradient scale -- only the transport differs from an 8-GPU node.

Checks (f32 mode, dropout 0, 3-task MTL step, global batch = 2 x B sequences per task, rank r holds sequences
[r B, (r + 1) B) of every task):
  * exact_graph_ln = True: averaged objective, averaged gradient and the parameters after ``--steps`` optimizer steps equal
    those of ONE process stepping on the global batch (the reference's semantics at that batch size: its graph-mode
    LayerNorm spans the whole batch, models/graph.py:43) up to f32 summation order;
  * exact_graph_ln = False (default, per-rank statistics): the same comparison differs by orders of magnitude more -- the
    mode is what makes the two agree, and the default is each replica = the reference at its LOCAL batch size;
  * both ranks end with bit-identical parameters in both modes;
  * bf16 mode (the benchmark's): the step captured on both ranks as STAGED hipGraphs (gradient exchange between the graph
    launches) and replayed gives the parameters of the eagerly issued steps, and both ranks stay bit-identical;
  * the sharded update (GradSync(shard_update=True): reduce-scatter -> Adam on the own half -> all-gather of the parameters)
    gives the parameters of the all-reduce path, and both ranks stay bit-identical.
The parent process never touches the GPU: it starts the two rank processes and relays rank 0's verdict (last stdout line,
JSON) and exit code.  Usage: python tools/two_rank_check.py [--hidden 512] [--batch 8] [--T 16] [--steps 2]"""
import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
ORDER = ("ar", "lta", "pnr")
F_IN, S, HEADS = 1536, 3, (115, 478)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--batch", type=int, default=8, help="sequences per task PER RANK")
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--port", type=int, default=0, help="rendezvous port (0: a free one picked by the parent)")
    ap.add_argument("--worker", type=int, default=-1)
    return ap.parse_args()


def build(args, device, seq_lo, seq_hi, sync):
    """Model, heads, MTLStep and the device batches of sequences [seq_lo, seq_hi) of every task of the GLOBAL dataset."""
    import torch
    from egopack_amd import data as D
    from egopack_amd import engine
    from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, PNRTask, RecognitionTask
    from egopack_amd.optim import FlatAdam

    torch.manual_seed(1)
    H = args.hidden
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": 0.0, "hidden_size": H}
    model = Graph(F_IN, hidden_size=H, depth=3, pre_dropout=0, temporal_pooling=trn, num_segments=S).to(device)
    tasks = {"ar": RecognitionTask(H, H, HEADS), "lta": LTATask(H, H, HEADS), "pnr": PNRTask(H, H)}
    for t in tasks.values():
        t.to(device)

    class DS:
        has_joint_label, num_labels = False, 2
    crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), DS()), "lta": MetricSelectorWrapper(CrossEntropyNone(), DS()),
            "pnr": BCEWithLogitsNone()}
    weights = {"ar": 1.0, "lta": 1.0, "pnr": 1.0, "oscc": 0.0}
    G = 2 * args.batch  # global sequences per task
    host, xs = {}, []
    for i, t in enumerate(ORDER):
        ds = D.SyntheticTaskDataset(t, G, args.T, S, 8, HEADS, k=1, seed=11)
        host[t] = D.collate([ds[j] for j in range(seq_lo, seq_hi)])
        gen = torch.Generator()
        gen.manual_seed(100 + i)
        x = torch.randn(G * args.T, S, F_IN, generator=gen)  # the global feature block of the task; this rank's rows
        xs.append(x[seq_lo * args.T: seq_hi * args.T])
    from egopack_amd import ops
    x_all = torch.cat(xs).to(device).to(ops.act_dtype())
    n = (seq_hi - seq_lo) * args.T
    dev = {}
    for i, t in enumerate(ORDER):
        b = host[t]
        b.x = torch.empty(0)
        d = b.to(device)
        d.x = x_all[i * n:(i + 1) * n]
        dev[t] = d
    merged = D.merge_batches([host[t] for t in ORDER]).to(device)
    merged.x = x_all
    params = [*model.parameters(), *(p for t in ORDER for p in tasks[t].parameters())]
    opt = FlatAdam(params, lr=1e-3, weight_decay=1e-5)
    step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=True, sync=sync)
    model.train()
    for t in tasks.values():
        t.train()
    return step, opt, dev, merged


def run(args, device, seq_lo, seq_hi, sync, exact):
    """(objective of step 1, gradient of step 1 [summed over ranks / world], parameters after ``steps`` steps)."""
    import torch
    from egopack_amd import dist as edist
    from egopack_amd import ops
    ops.manual_seed(5)
    step, opt, dev, merged = build(args, device, seq_lo, seq_hi, sync)
    step.exact_graph_ln = exact
    step.use_graph = False
    total, _ = step.forward_backward(dev, merged)
    if not opt.materialised:
        opt._materialise()  # the flat buffers (the first optimizer step builds them from the parameters' gradients)
    grad = opt.flat_g.detach().clone()
    obj = total.detach().double().reshape(1).clone()
    if sync is not None:
        edist.all_reduce_sum_(grad, sync.group)
        grad /= sync.world
        edist.all_reduce_sum_(obj, sync.group)
        obj /= sync.world
    step._exchange_and_update()
    for _ in range(args.steps - 1):
        step.step(dev, merged)
    torch.cuda.synchronize()
    return float(obj.item()), grad.cpu(), opt.flat_p.detach().cpu().clone(), opt


def run_replayed(args, device, seq_lo, seq_hi, sync, graph: bool):
    """Parameters after 4 optimizer steps: eagerly, or 2 eager warm-up steps + the captured step replayed twice."""
    import torch
    from egopack_amd import ops
    ops.manual_seed(5)
    step, opt, dev, merged = build(args, device, seq_lo, seq_hi, sync)
    if graph:
        step.capture(dev, merged, warmup=2)
        kind = ("staged graphs" if isinstance(step._graph, list) else
                "one graph incl. the gradient exchange" if getattr(step, "_graph_has_exchange", False) else "one graph")
        for _ in range(2):
            step.replay()
    else:
        kind = "eager"
        for _ in range(4):
            step.step(dev, merged)
    torch.cuda.synchronize()
    return opt.flat_p.detach().cpu().clone(), kind


def worker(args):
    import torch
    import torch.distributed as dist
    from egopack_amd import dist as edist
    from egopack_amd import ops
    rank = args.worker
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(args.port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    device = torch.device("cuda:0")
    ops.set_compute("f32")
    B = args.batch
    out = {}
    res = {}
    for exact in (True, False):
        sync = edist.GradSync(2)
        obj, grad, par, opt = run(args, device, rank * B, (rank + 1) * B, sync, exact)
        other = [torch.empty_like(par) for _ in range(2)]
        dist.all_gather(other, par)
        res[exact] = (obj, grad, par, bool(torch.equal(other[0], other[1])), opt)
    # the benchmark's mode: bf16, the step captured as staged hipGraphs with the exchange between them, two ranks
    ops.set_compute("bf16")
    rep = {}
    for graph in (False, True):
        par, kind = run_replayed(args, device, rank * B, (rank + 1) * B, edist.GradSync(2), graph)
        other = [torch.empty_like(par) for _ in range(2)]
        dist.all_gather(other, par)
        rep[graph] = (par, kind, bool(torch.equal(other[0], other[1])))
    # sharded update (reduce-scatter -> Adam on the own half -> all-gather of the parameters), captured step replayed
    par_s, kind_s = run_replayed(args, device, rank * B, (rank + 1) * B, edist.GradSync(2, shard_update=True), True)
    other = [torch.empty_like(par_s) for _ in range(2)]
    dist.all_gather(other, par_s)
    shard = (par_s, kind_s, bool(torch.equal(other[0], other[1])))
    ops.set_compute("f32")
    dist.barrier()
    if rank == 0:
        obj1, grad1, par1, opt1 = run(args, device, 0, 2 * B, None, False)  # ONE process on the global batch

        def rel(a, b):
            return float((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-30))
        for exact in (True, False):
            obj, grad, par, same, _ = res[exact]
            k = "exact" if exact else "local"
            out[k] = {"objective": obj, "objective_rel": abs(obj - obj1) / abs(obj1), "grad_rel": rel(grad, grad1),
                      "param_frac_within_2e-4": float(((par - par1).abs() <= 2e-4).double().mean()),
                      "param_max_abs": float((par - par1).abs().max()), "ranks_bit_identical": same}
        out["objective_one_process"] = obj1
        pe, pg = rep[False][0], rep[True][0]
        out["bf16_replay"] = {"capture": rep[True][1], "ranks_bit_identical": rep[True][2] and rep[False][2],
                              "replayed_vs_eager_rel": float((pg.double() - pe.double()).norm() / pe.double().norm()),
                              "replayed_vs_eager_max_abs": float((pg - pe).abs().max())}
        out["sharded_update"] = {"capture": shard[1], "ranks_bit_identical": shard[2],
                                 "vs_allreduce_replay_max_abs": float((shard[0] - pg).abs().max())}
        if os.environ.get("EGK_DBG_SHARD"):
            bad = ((shard[0] - pg).abs() > 1e-6).nonzero().flatten()
            print("[dbg] n", pg.numel(), "bad", bad.numel(), "first", bad[:5].tolist(), "last", bad[-5:].tolist(), file=sys.stderr, flush=True)
            runs, prev, start = [], None, None
            for i in bad.tolist():
                if prev is None or i != prev + 1:
                    if start is not None:
                        runs.append((start, prev + 1))
                    start = i
                prev = i
            if start is not None:
                runs.append((start, prev + 1))
            print("[dbg] runs", runs[:20], len(runs), file=sys.stderr, flush=True)
        out["config"] = dict(hidden=args.hidden, batch_per_rank=B, T=args.T, steps=args.steps, mode="f32", tasks=list(ORDER))
        e, l = out["exact"], out["local"]
        ok = (e["objective_rel"] <= 1e-6 and e["grad_rel"] <= 2e-3 and e["param_frac_within_2e-4"] >= 0.999
              and e["ranks_bit_identical"] and l["ranks_bit_identical"]
              and l["grad_rel"] >= 100 * e["grad_rel"] and l["param_frac_within_2e-4"] < 0.99
              and out["bf16_replay"]["capture"] == "staged graphs" and out["bf16_replay"]["ranks_bit_identical"]
              and out["bf16_replay"]["replayed_vs_eager_rel"] <= 1e-6
              and out["sharded_update"]["ranks_bit_identical"] and out["sharded_update"]["vs_allreduce_replay_max_abs"] <= 1e-6)
        out["ok"] = bool(ok)
        print(json.dumps(out), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return 0 if ok else 1
    dist.barrier()
    dist.destroy_process_group()
    return 0


def main():
    args = parse()
    if args.worker >= 0:
        sys.exit(worker(args))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not args.port:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            args.port = sk.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(Path(__file__).resolve()), "--worker", str(r), "--hidden", str(args.hidden),
                               "--batch", str(args.batch), "--T", str(args.T), "--steps", str(args.steps), "--port", str(args.port)],
                              env=env, stdout=subprocess.PIPE if r == 0 else None, text=True) for r in range(2)]
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode, procs[1].wait()]
    sys.stdout.write(out0)
    sys.exit(0 if rcs == [0, 0] else 1)


if __name__ == "__main__":
    main()
