#!/usr/bin/env python3
"""bench.py -- clip-seqs/sec of the EgoPack multi-task training step on MI355X.

Workload (BASELINE.json metric / configs[2] at N GPUs): AR + LTA + PNR multi-task
pre-training, per GPU and per task B=64 sequences of T=32 clip nodes, 3 x 1536-d Omnivore-shaped
features per node, hidden 1024, TRN hidden 1024 (dropout 0.5), backbone depth 3, temporal radius
k=1, Adam.  A step = zero_grad -> fused backbone forward over the three task batches -> heads ->
sum_t w_t * loss_t.mean() -> backward -> (gradient all-reduce) -> Adam, inputs resident in HBM.
One process per GPU (torchrun env), weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0 (contract in the task statement) with ``roofline`` (dominant kernel,
HIP-event timings taken by the library on the launch stream) and ``cpu_baseline`` (the CPU oracle
timed on the host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

import torch  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA TFLOP/s (MI355X_MICROARCH.md: chip-level parameters)
PEAK_HBM_GBS = 8000.0


WORKLOADS = {
    # name: (enabled tasks, note).  "mtl" is the headline (BASELINE.json metric / configs[2]); the others are the
    # remaining BASELINE configs, selectable for evidence but never the default.
    "mtl": (("ar", "lta", "pnr"), "MTL pre-train AR+LTA+PNR (BASELINE config 3)"),
    "ar": (("ar",), "GraphONE-free temporal backbone, AR single task (BASELINE config 2)"),
    "mtl4": (("ar", "lta", "oscc", "pnr"), "4-task MTL (BASELINE config 5: use --T 256 --batch 16)"),
    "egopack_oscc": (("oscc",), "EgoPack novel task OSCC: frozen AR/LTA/PNR prototype banks + GraphONE (BASELINE config 4)"),
}


def build_workload(args, rank, device):
    from egopack_amd import data as D
    from egopack_amd.criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
    from egopack_amd.models import Graph
    from egopack_amd.models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask

    torch.manual_seed(1)  # identical initial parameters on every rank (defaults.yaml:2)
    H, HP, F_IN, S, heads = args.hidden, args.trn_hidden, 1536, 3, (115, 478)
    trn = {"_target_": "egopack_amd.models.temporal_pooling.trn_pooling.TRNPooling", "dropout": args.dropout,
           "hidden_size": HP}
    model = Graph(F_IN, hidden_size=H, depth=3, pre_dropout=0, temporal_pooling=trn, num_segments=S)
    tasks = {"ar": RecognitionTask(H, H, heads), "oscc": OSCCTask(H, H), "lta": LTATask(H, H, heads), "pnr": PNRTask(H, H)}

    class DS:
        has_joint_label, num_labels = False, 2
    crit = {"ar": MetricSelectorWrapper(CrossEntropyNone(), DS()), "lta": MetricSelectorWrapper(CrossEntropyNone(), DS()),
            "oscc": CrossEntropyNone(), "pnr": BCEWithLogitsNone()}
    order = WORKLOADS[args.workload][0]
    weights = {t: (1.0 if t in order else 0.0) for t in ("ar", "oscc", "lta", "pnr")}

    # synthetic batches: labels / positions / edges from the host-side dataset logic, features N(0,1)
    # generated straight into HBM (seed 1 + rank: SURVEY 8d)
    host = {}
    for t in order:
        ds = D.SyntheticTaskDataset(t, args.batch, args.T, S, 8, heads, k=1, seed=1 + rank)  # tiny x, replaced below
        host[t] = D.collate([ds[i] for i in range(args.batch)])
    gen = torch.Generator(device=device)
    gen.manual_seed(1 + rank)
    n = args.batch * args.T
    # one resident feature buffer for the step (what data.pack_features builds from loader batches); the
    # task batches are row ranges of it.  bf16 storage for the bf16 configs.
    x_all = torch.randn(len(order) * n, S, F_IN, device=device, generator=gen)
    if args.compute == "bf16":
        x_all = x_all.to(torch.bfloat16)
    dev = {}
    for i, t in enumerate(order):
        b = host[t]
        b.x = torch.empty(0)
        d = b.to(device)
        d.x = x_all[i * n:(i + 1) * n]
        dev[t] = d
    merged = None
    if len(order) > 1:
        merged = D.merge_batches([host[t] for t in order]).to(device)
        merged.x = x_all
    if args.workload == "egopack_oscc":  # aux-classifier heads + frozen prototype banks
        aux = {"ar": ("oscc", "lta", "pnr"), "oscc": ("ar", "lta", "pnr"), "lta": ("ar", "oscc", "pnr"), "pnr": ("ar", "oscc", "lta")}
        tasks = {"ar": RecognitionTask(H, H, heads, aux_tasks=aux["ar"]), "oscc": OSCCTask(H, H, aux_tasks=aux["oscc"], average_logits=True),
                 "lta": LTATask(H, H, heads, aux_tasks=aux["lta"]), "pnr": PNRTask(H, H, aux_tasks=aux["pnr"])}
    return model, tasks, crit, weights, dev, merged


def usable_cores() -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(sds, names, dev, weights, sample_batch=None, budget_s=20.0, egopack=None):
    """The CPU oracle (oracle/path.py, checker code) timed on the host cores on a BOUNDED sample of the
    same workload: the same step (fp32, forward + backward + torch.optim.Adam, same tasks, same T and
    model) on the FULL batch of B sequences per task (graph-LayerNorm couples the samples of a batch:
    a step on fewer sequences is a different computation), as many steps as fit the budget (>= 1, at
    most 20).  ``sample_batch`` < B restricts it to the first sequences (development only)."""
    from oracle import path as O
    from oracle import pyg_ops as P
    cores = min(usable_cores(), 64)
    torch.set_num_threads(cores)
    leaf = {g: {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and not k.endswith("frequency")
                                                      and not k.startswith("embeddings.")) else v.clone())
                for k, v in sd.items()} for g, sd in sds.items()}  # (GraphONE's prototype banks are frozen: graphONE.py:48)
    batches = {}
    for t, d in dev.items():
        B, n = d.num_graphs, d.pos.shape[0]
        T = n // B
        b = B if sample_batch is None else min(sample_batch, B)
        rows = b * T  # the first b sequences (collation keeps sequences contiguous)
        ei = d.edge_index.cpu()
        ei = ei[:, (ei[0] < rows) & (ei[1] < rows)]
        y = d.y.cpu()
        batches[t] = P.OData(x=d.x[:rows].float().cpu(), pos=d.pos[:rows].cpu(), edge_index=ei, batch=d.batch[:rows].cpu(),
                             y=y[:b] if y.shape[0] == B else y[:rows], num_graphs=b)
    flat = [p for g in leaf.values() for p in g.values() if p.requires_grad]
    opt = torch.optim.Adam(flat, lr=1e-5, weight_decay=1e-5)
    seqs = sum(b.num_graphs for b in batches.values())

    def one():
        opt.zero_grad()
        if egopack is not None:
            # the novel-task step of BASELINE config 4 (reference main_egopack.py:45-61,64-159): backbone forward, the detached
            # auxiliary projections -> GraphONE over the frozen banks -> fused logits -> the primary task's loss; backward; Adam
            d = batches[egopack["primary"]]
            feat = O.graph_forward(leaf["temporal_graph"], d.x, d.pos, d.edge_index, 3)
            loss, _, _, _ = O.egopack_task_loss(egopack["primary"], {t: leaf[n] for t, n in names.items()}, leaf["graphone"], feat,
                                                d.batch, d.y, egopack["others"], egopack["k"], egopack["depth"], True, True,
                                                num_graphs=d.num_graphs)
            total = weights[egopack["primary"]] * loss.mean()
        else:
            total, _ = O.mtl_objective(leaf["temporal_graph"], {t: leaf[n] for t, n in names.items()}, batches, weights)
        total.backward()
        opt.step()

    t0 = time.perf_counter()
    one()  # warm-up (also sizes the sample)
    first = time.perf_counter() - t0
    n = max(1, min(20, int(budget_s / max(first, 1e-3)) - 1)) if first < budget_s else 0
    if n:
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        dt = (time.perf_counter() - t0) / n
    else:
        dt = first
    t0_ = next(iter(batches))
    b0 = batches[t0_]
    return {"value": seqs / dt, "unit": "clip-seqs/s", "cores": cores, "kind": "port",
            "sample": f"{n or 1} step(s){' after 1 warm-up' if n else ' (the first one)'} of the same step on "
                      f"{b0.num_graphs} of the {dev[t0_].num_graphs} sequences per task ({len(batches)} task(s) x T="
                      f"{b0.x.shape[0] // b0.num_graphs}), fp32 torch CPU oracle, {cores} threads, {dt * 1e3:.0f} ms/step"}


# library profiler name -> kernel symbol (substring) in rocprofv3 output
def _pipe(ta, tb, kg, ni=4):
    return f"gemm_pipe_kernel<2, {ta}, {tb}, {kg}, 1, {ni}"  # (a prefix: later template arguments vary)


ROCPROF_NAME = {"gemm_bf16_nn": _pipe("false", "false", 1), "gemm_bf16_nt": _pipe("false", "true", 1),
                "gemm_bf16_tt": _pipe("true", "true", 1), "gemm_bf16_tn": _pipe("true", "false", 1),
                "gemm_bf16_nn_g2": _pipe("false", "false", 2), "gemm_bf16_nt_g2": _pipe("false", "true", 2),
                "gemm_bf16_tt_g2": _pipe("true", "true", 2), "gemm_bf16_tn_g2": _pipe("true", "false", 2),
                "gemm_bf16_nn_r96": _pipe("false", "false", 1, 3), "gemm_bf16_nt_r96": _pipe("false", "true", 1, 3),
                "gemm_bf16_nn_r64": _pipe("false", "false", 1, 2), "gemm_bf16_nt_r64": _pipe("false", "true", 1, 2),
                "gemm_bf16_nn_r192": "gemm_pipe_kernel<3, false, false, 1, 2, 3", "gemm_bf16_nt_r192": "gemm_pipe_kernel<3, false, true, 1, 2, 3",
                "gemm_bf16_nn_t256": "gemm_big_kernel<false, false", "gemm_bf16_nt_t256": "gemm_big_kernel<false, true",
                "gemm_bf16_tt_t256": "gemm_big_kernel<true, true", "gemm_bf16_tn_t256": "gemm_big_kernel<true, false",
                "gemm_bf16_group_nn": "gemm_pipe_group_kernel<2, false, false", "gemm_bf16_group_nt": "gemm_pipe_group_kernel<2, false, true",
                # (the weight-gradient groups run on two instantiations: 128 x 128 tiles, and 256 x 128 tiles where those load the CUs unevenly)
                "gemm_bf16_group_tt": "gemm_pipe_group_kernel<2, true, true | gemm_pipe_group_kernel<3, true, true",
                "gemm_bf16_generic": "gemm_kernel<true", "gemm_splitk_reduce": "gemm_splitk_reduce",
                "adam": "adam_kernel<", "csr_gather": "csr_gather_"}


def pmc_key(args) -> str:
    """Name of the configuration a PMC summary belongs to (profiles/pmc_<key>.json, written by tools/stamp_profile.py <tag> <key>):
    the same kernel name at other sizes moves other bytes, so every benchmarked configuration has a file of its own."""
    return f"{args.workload}_B{args.batch}_T{args.T}_H{args.hidden}_Hp{args.trn_hidden}_{args.compute}"


def pmc_file(args=None) -> Path:
    if args is not None:
        f = REPO / "profiles" / f"pmc_{pmc_key(args)}.json"
        if f.exists():
            return f
        if pmc_key(args) != "mtl_B64_T32_H1024_Hp1024_bf16":
            return f  # (absent: traffic stays null for this configuration)
    return REPO / "profiles" / "pmc_latest.json"  # the default workload's passes (rounds 1-3 wrote this name)


def pmc_traffic(kernel: str, args=None):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes of THIS configuration (profiles/pmc_<key>.json,
    written by tools/profile.sh + tools/stamp_profile.py: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench,
    FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).  None when no profile of the configuration is committed."""
    f = pmc_file(args)
    sym = ROCPROF_NAME.get(kernel)
    if not f.exists() or sym is None:
        return None
    try:
        pmc = json.loads(f.read_text())
        # several instantiations can share the live timer's name (e.g. the grouped launch with bf16 and with f16 operands): the
        # launch-weighted mean per counter over all of them, then the sum of the counters
        byts, launches = {}, {}
        for name, ctrs in pmc.items():
            if name.startswith("_") or not any(part in name for part in sym.split(" | ")):
                continue
            for ctr, c in ctrs.items():
                byts[ctr] = byts.get(ctr, 0.0) + c["bytes_per_launch"] * c["launches"]
                launches[ctr] = launches.get(ctr, 0) + c["launches"]
        return sum(byts[c] / launches[c] for c in byts if launches[c]) if byts else None
    except Exception:
        return None


def pmc_provenance(args=None) -> str:
    """Where ``roofline.traffic`` comes from: the committed PMC summary and the source commit it was measured at
    (``_meta`` of the file, written by tools/stamp_profile.py when the summary is copied into profiles/)."""
    f = pmc_file(args)
    rel = f"profiles/{f.name}"
    try:
        meta = json.loads(f.read_text()).get("_meta", {})
    except Exception:  # noqa: BLE001
        return f"{rel} (absent)"
    return (f"{rel}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench (tools/profile.sh), "
            f"FETCH x2 per the gfx950 correction; measured at source commit {meta.get('commit', 'unknown')} "
            f"({meta.get('tag', '?')}, {meta.get('date', '?')}); NOT re-measured in this run")


def roofline(ops, step_fn, compute, n_steps=3, args=None):
    """Profile ``n_steps`` eager steps with the library's HIP-event timers; report the dominant kernel."""
    ops.prof_reset()
    ops.prof_enable(True)
    # Keep the device queue FULL while the eager steps are issued: the start event of a launch is stamped when the
    # command processor reaches it, so on an empty queue (the host-bound forward pass) the host's time between
    # hipEventRecord and the kernel launch call (5-9 us) would be counted as kernel time -- the forward contraction read
    # 42 us live against 33 us in the rocprofv3 trace of the same launches.  Behind a ~40 ms spin the three steps are
    # enqueued ahead of the device and every launch is timed back to back, the way the captured graph runs them.
    torch.cuda._sleep(int(8e7))
    for _ in range(n_steps):
        step_fn()
    torch.cuda.synchronize()
    ops.prof_enable(False)
    rep = ops.prof_report()
    ops.prof_reset()
    if not rep:
        return None, {}
    name, r = max(rep.items(), key=lambda kv: kv[1]["total_ms"])
    avg_ms = r["total_ms"] / r["launches"]
    if name.startswith("gemm"):
        achieved = r["flops"] / (r["total_ms"] * 1e-3) / 1e12
        peak = PEAK["bf16" if "bf16" in name else "f32"]
        out = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak}
    else:
        achieved = r["bytes"] / (r["total_ms"] * 1e-3) / 1e9
        out = {"bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS}
    out.update({"traffic": pmc_traffic(name, args), "traffic_source": pmc_provenance(args),
                "kernel": name, "rocprof_symbol": ROCPROF_NAME.get(name), "launches_per_step": r["launches"] / n_steps, "avg_launch_us": avg_ms * 1e3,
                "alg_per_launch": (r["flops"] if name.startswith("gemm") else r["bytes"]) / r["launches"]})
    # whole step against the same peaks: the sum over every launch of max(algorithmic flops / MFMA peak,
    # algorithmic bytes / HBM peak) -- what the step would take if each kernel ran at its bounding roofline with no
    # launch gaps; main() divides it by the measured step
    lb = 0.0
    for k, v in rep.items():
        t = v["bytes"] / (PEAK_HBM_GBS * 1e9)
        if k.startswith("gemm") and v["flops"]:
            t = max(t, v["flops"] / (PEAK["bf16" if "bf16" in k else "f32"] * 1e12))
        lb += t
    out["step"] = {"lower_bound_ms": lb / n_steps * 1e3, "flops_per_step": sum(v["flops"] for v in rep.values()) / n_steps,
                   "bytes_per_step": sum(v["bytes"] for v in rep.values()) / n_steps}
    table = {k: {"launches_per_step": v["launches"] / n_steps, "ms_per_step": v["total_ms"] / n_steps,
                 "tflops": (v["flops"] / (v["total_ms"] * 1e-3) / 1e12) if v["flops"] and v["total_ms"] else None,
                 "gbs": (v["bytes"] / (v["total_ms"] * 1e-3) / 1e9) if v["bytes"] and v["total_ms"] else None}
             for k, v in sorted(rep.items(), key=lambda kv: -kv[1]["total_ms"])}
    return out, table


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64, help="sequences per task per GPU")
    ap.add_argument("--T", type=int, default=32, help="clip nodes per sequence")
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--trn-hidden", type=int, default=1024)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--compute", choices=["bf16", "bf16_f32act", "bf16x3", "f32"], default="bf16")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="mtl")
    ap.add_argument("--bank", type=int, default=4096, help="prototypes per task bank (egopack_oscc)")
    ap.add_argument("--graphone-k", type=int, default=4)
    ap.add_argument("--graphone-depth", type=int, default=3)
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph")
    ap.add_argument("--no-fused-backbone", action="store_true")
    ap.add_argument("--serial-heads", action="store_true", help="run the task heads on the main stream")
    ap.add_argument("--feature-store", type=int, default=0, metavar="ROWS",
                    help="assemble every step's input block inside the step from a device-resident feature store of ROWS rows "
                         "(egk_gather_rows on a fixed index matrix, captured with the step): the input-pipeline-inclusive rate")
    ap.add_argument("--exchange-dry-run", type=int, default=0, metavar="N",
                    help="one GPU: run the N-rank gradient-exchange path (staged backward, conversion, RCCL all-reduce on a "
                         "1-rank group, per-chunk Adam) -- everything of the N-GPU step except the time on the xGMI links")
    ap.add_argument("--head-streams", type=int, default=0, help="cap on the HIP streams the task heads are spread over (0 = one per task)")
    ap.add_argument("--staged", choices=["auto", "on", "off"], default="auto",
                    help="three-stage backward with region-wise gradient exchange (auto: when there are several ranks)")
    ap.add_argument("--no-wgrad-streams", action="store_true", help="keep the weight-gradient launches on the backward stream")
    ap.add_argument("--one-call-backward", action="store_true",
                    help="one backward() call over all head streams instead of one per head inside its stream context (A/B)")
    ap.add_argument("--grad-compress", choices=["bf16", "none"], default="none",
                    help="element type of the gradient all-reduce when --gpus > 1: 'none' = the f32 exchange the entry "
                         "points train with (main_temporal.py grad_compress default); 'bf16' = the half-size variant")
    ap.add_argument("--one-gpu-gloo", action="store_true",
                    help="development: run the N ranks on ONE GPU (every rank on cuda:0) over a gloo group -- the N-rank step, "
                         "capture mode and timing protocol on a one-GPU box (RCCL refuses two ranks on one device); the "
                         "throughput it prints is NOT a multi-GPU figure and says so in config.transport")
    ap.add_argument("--exchange-graph", choices=["auto", "one", "staged"], default="auto",
                    help="N ranks on an RCCL group: 'staged' = three hipGraphs with the collectives issued eagerly between the graph "
                         "launches (the default path: two real rank processes have run it); 'one' = ONE hipGraph that also holds the "
                         "collectives and the per-chunk Adam launches (faster, has only met a 1-rank group); 'auto' = 'one' if a "
                         "pre-flight CHILD process per rank captured and replayed it with exit code 0, else 'staged'")
    ap.add_argument("--probe-timeout", type=float, default=240.0, help="seconds the pre-flight children of --exchange-graph auto may take")
    ap.add_argument("--probe-child", action="store_true", help=argparse.SUPPRESS)  # (this process IS a pre-flight child)
    ap.add_argument("--strict-capture", action="store_true",
                    help="fail instead of falling back (staged graphs -> one-piece graph -> eager) when a capture fails")
    ap.add_argument("--gemm-knob", type=str, default=None, help="development: value(s, comma list) passed to egk_gemm_set_pipeline before the run (A/B on one box)")
    ap.add_argument("--no-early-adam", action="store_true", help="A/B: one Adam launch after the whole backward")
    ap.add_argument("--force-wgrad-streams", action="store_true", help="A/B: weight gradients on a side stream also for single-task steps")
    ap.add_argument("--egk-tune", default="", help="development: comma list key=value passed to egk_tune (row-kernel grid caps)")
    ap.add_argument("--hw-queues", type=int, default=0,
                    help="GPU_MAX_HW_QUEUES for this process (0 = runtime default, 4).  3 measured 1.3 %% faster on the three-head "
                         "step but crashed hipGraphLaunch in other stream configurations: opt-in only")
    ap.add_argument("--last-wgrad-side", action="store_true", help="A/B: the last weight gradient goes to the side stream like the others")
    ap.add_argument("--ln-reduce-inline", action="store_true", help="A/B: the norm layers' dw / db reduction stays on the backward stream")
    ap.add_argument("--no-classifier-bank", action="store_true", help="A/B: one contraction per classifier instead of one per head")
    ap.add_argument("--csr-split-heavy", action="store_true",
                    help="A/B: sum the listed heavy CSR rows with the split launches even when they are short enough for the launch itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-f32-leg", action="store_true",
                    help="skip the short reference-precision (--compute f32) measurement added to the JSON line at N = 1")
    ap.add_argument("--f32-steps", type=int, default=50, help="timed steps of the reference-precision leg")
    ap.add_argument("--min-timed-s", type=float, default=0.5,
                    help="repeat the timed K-step block until at least this much stepping has been timed; the median block is reported")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table (stderr)")
    ap.add_argument("--stamps", action="store_true",
                    help="development: capture one-lane wall-clock stamps at the phase boundaries of the step and print the "
                         "phase table of the last replay (stderr); each stamp costs ~2 us of the step")
    ap.add_argument("--no-grouped-heads", action="store_true", help="A/B: one projection chain per task instead of grouped launches")
    ap.add_argument("--no-wgrad-grouping", action="store_true", help="A/B: every weight gradient as its own (split-K) launch")
    return ap.parse_args(argv)


def spawn_plan(n: int, argv, port: int = None):
    """Command that starts ``n`` ranks of this script on this node -- exactly what the driver runs for N > 1:
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>``."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), str(Path(__file__).resolve()), *argv]


def spawn_ranks(n: int, argv) -> int:
    """``python bench.py --gpus N`` without a torchrun environment: start N ranks as CHILD processes (this parent has not
    touched the GPU and never does; nothing is exec'ed), relay their output, keep rank 0's JSON line as the last line of
    stdout, and return non-zero if any rank failed."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC only on this pool: RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // max(n, 1))))
    proc = subprocess.run(spawn_plan(n, argv), env=env, stdout=subprocess.PIPE, text=True)
    lines = proc.stdout.splitlines()
    result = next((ln for ln in reversed(lines) if ln.startswith("{") and '"metric"' in ln), None)
    for ln in lines:
        if ln is not result:
            print(ln)
    if result is not None:
        print(result, flush=True)
    if proc.returncode != 0 or result is None:
        print(f"[bench] {n}-rank run failed (exit code {proc.returncode}, result line {'present' if result else 'missing'})",
              file=sys.stderr, flush=True)
        return proc.returncode or 1
    return 0


def one_graph_probe(args, argv, rank: int, world: int):
    """Pre-flight for ``--exchange-graph auto``: every rank starts a CHILD process (this rank process has not touched the GPU
    yet) that runs the N-rank step as ONE hipGraph incl. the RCCL collectives -- capture, replays, the exchange-cost leg's
    re-captures -- over its own rendezvous (a free port rank 0 picked), and exits 0 only if all of it worked and the ranks
    ended bit-identical.  The ranks poll their children in lock step over the gloo coordination group: one failed / timed-out
    child anywhere sends EVERY rank to the staged graphs (the modes issue different collectives), the remaining children are
    killed, and the run goes on.  -> (ok, note)."""
    import subprocess
    import tempfile
    import torch.distributed as dist
    from egopack_amd.dist import free_port
    port = [free_port() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(port, src=0)
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port[0]))
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (the child ranks rendezvous among themselves: rank 0's child hosts the store)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, str(Path(__file__).resolve()), *argv, "--probe-child", "--exchange-graph", "one", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-roofline", "--no-f32-leg", "--strict-capture", "--min-timed-s", "0"]
    if os.environ.get("EGK_TEST_PROBE_CMD"):  # (tests of the lock-step polling itself: a stand-in child, no GPU involved)
        import shlex
        cmd = shlex.split(os.environ["EGK_TEST_PROBE_CMD"])
    log = tempfile.NamedTemporaryFile("w+", prefix=f"egk_probe_r{rank}_", suffix=".log", delete=False)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=log)
    deadline = time.monotonic() + args.probe_timeout
    inject = os.environ.get("EGK_TEST_KILL_PROBE") == str(rank)  # (tests: the probe child of this rank is killed mid-way)
    t_inject = time.monotonic() + 2.0
    note = ""
    while True:
        rc = proc.poll()
        if rc is None and inject and time.monotonic() > t_inject:
            proc.kill()
            rc = proc.wait()
        if rc is None and time.monotonic() > deadline:
            proc.kill()
            rc = proc.wait()
            note = f"timed out after {args.probe_timeout:.0f} s"
        state = [0, 1] if rc is None else ([1, 0] if rc != 0 else [0, 0])  # [failed, running]
        flags = torch.tensor(state, dtype=torch.int32)
        if world > 1:
            dist.all_reduce(flags)
        if int(flags[0]):  # a child failed somewhere: nobody waits for the others
            if proc.poll() is None:
                proc.kill()
                proc.wait()
            if rc is not None and rc != 0:
                log.flush()
                tail = Path(log.name).read_text()[-1500:]
                note = note or f"child exit code {rc}"
                print(f"[bench] rank {rank}: one-graph probe child failed ({note}); its stderr ends:\n{tail}", file=sys.stderr, flush=True)
            return False, (note or "failed on another rank")
        if not int(flags[1]):
            return True, "passed"
        time.sleep(0.25)


def measure(args, rank, world, device, steps, warmup, want_roofline=True, want_cpu=True, min_timed_s=0.0, group=None):
    """Build the workload in ``args.compute`` mode, capture / warm up, time ``steps`` steps (barrier + synchronize on both
    sides, MAX over ranks) and take the roofline / CPU-baseline legs.  Returns a dict of results."""
    from egopack_amd import dist as edist
    from egopack_amd import engine, ops
    from egopack_amd.optim import FlatAdam
    ops.set_compute(args.compute)
    ops.manual_seed(1000 + rank)  # dropout streams differ per rank
    if getattr(args, "stamps", False):
        ops.stamps_enable(device)
    model, tasks, crit, weights, dev, merged = build_workload(args, rank, device)
    names = {"ar": "task/recognition", "oscc": "task/oscc", "lta": "task/lta", "pnr": "task/pnr"}
    sds = None
    if rank == 0 and world == 1 and want_cpu and not args.no_cpu_baseline and args.workload in ("mtl", "ar", "mtl4", "egopack_oscc"):
        sds = {"temporal_graph": {k: v.clone() for k, v in model.state_dict().items()}}
        for t, n in names.items():
            sds[n] = {k: v.clone() for k, v in tasks[t].state_dict().items()}
    if args.no_classifier_bank:
        for t in tasks.values():
            for p in t.parameters():
                if hasattr(p, "_egk_bank"):
                    del p._egk_bank
    model.to(device).train()
    for t in tasks.values():
        t.to(device).train()
    params = [*model.parameters(), *(p for t in tasks.values() for p in t.parameters())]
    sync = edist.GradSync(world, compress=args.grad_compress, group=group) if world > 1 else None
    if world == 1 and args.exchange_dry_run > 1:
        edist.init_single_rank_group()  # (free rendezvous port)
        sync = edist.GradSync(args.exchange_dry_run, compress=args.grad_compress)
    # coordination (flags, timings, barriers) goes over the DEFAULT group: gloo when this process was set up by main()
    # (CPU tensors), RCCL otherwise (device tensors)
    coord_cpu = world > 1 and torch.distributed.get_backend() == "gloo"
    fused_merged = None if args.no_fused_backbone else merged
    if args.workload == "egopack_oscc":
        from egopack_amd.models.graphONE.graphONE import GraphONE
        gen = torch.Generator(device=device)
        gen.manual_seed(7)
        banks = {t: torch.randn(args.bank, args.hidden, device=device, generator=gen) for t in ("ar", "lta", "pnr")}
        graphone = GraphONE(banks, features_size=args.hidden, hidden_size=args.hidden, k=args.graphone_k,
                            depth=args.graphone_depth, residual=True).to(device)
        params += list(graphone.parameters())
        if sds is not None:
            sds["graphone"] = {k: v.detach().cpu().clone() for k, v in graphone.state_dict().items()}
        opt = FlatAdam(params, lr=1e-5, weight_decay=1e-5)
        step = engine.EgoPackStep(model, tasks, graphone, weights, opt, backprop_temporal_graph=True,
                                  temporal_graph_train_mode=False, sync=sync)
        if args.no_wgrad_streams:
            step.wgrad_side_streams = False
        if args.force_wgrad_streams:
            step.wgrad_side_streams = True
    else:
        opt = FlatAdam(params, lr=1e-5, weight_decay=1e-5)  # defaults.yaml:17-20
        step = engine.MTLStep(model, tasks, crit, weights, opt, fused_backbone=not args.no_fused_backbone, sync=sync,
                              parallel_heads=not args.serial_heads)
        if args.no_wgrad_streams:
            step.wgrad_side_streams = False
        step.staged = {"auto": None, "on": True, "off": False}[args.staged]
        if args.head_streams:
            step.max_head_streams = args.head_streams
        if args.one_call_backward:
            step.headwise_backward = False
        if args.no_early_adam:
            step.early_adam = False
        if args.force_wgrad_streams:
            step.wgrad_side_streams = True
        if getattr(args, "no_grouped_heads", False):
            step.grouped_heads = False
    if getattr(args, "no_wgrad_grouping", False):
        step.wgrad_grouping = False
    step.one_graph_exchange = getattr(args, "exchange_graph", "staged") == "one"  # ('auto' was resolved by main(): probe)

    def eager_step():
        step.step(dev, fused_merged)

    if args.feature_store and fused_merged is not None and torch.is_tensor(fused_merged.x):
        from egopack_amd import feature_store as FS
        buf = fused_merged.x  # the packed [sum N, S, F] input block every task batch views
        store = FS.FeatureStore.__new__(FS.FeatureStore)
        store.table = torch.randn(args.feature_store, buf.shape[-1], device=device).to(buf.dtype)
        store.rows, store.features_size, store.offsets = args.feature_store, buf.shape[-1], {}
        gen = torch.Generator().manual_seed(3 + rank)
        idx = torch.randint(0, args.feature_store, buf.shape[:-1], generator=gen).to(device)
        step.input_hook = lambda: store.gather(idx, out=buf)

    # which execution mode actually ran is part of the result (config.capture): 'staged graphs' (three hipGraphs with the
    # gradient exchange between them), 'one graph', or 'eager'; a failed capture is reported, never silent
    capture, fallbacks = "eager", []

    def all_ranks_ok(ok: bool) -> bool:
        """A capture that fails on ONE rank must send EVERY rank to the next mode: the modes issue different collectives."""
        if world <= 1:
            return ok
        flag = torch.tensor([1 if ok else 0], device="cpu" if coord_cpu else device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item())

    if args.mode == "graph":
        run = None
        for attempt in ("as configured", "one-piece backward", "eager"):
            err = None
            try:
                if attempt == "one-piece backward":
                    step.staged = False
                if attempt == "eager":
                    eager_step()
                    run, capture = eager_step, "eager"
                else:
                    step.capture(dev, fused_merged, warmup=2)
                    run = step.replay
                    capture = ("staged graphs" if isinstance(step._graph, list) else
                               "one graph incl. the gradient exchange" if getattr(step, "_graph_has_exchange", False) else "one graph")
                    fallbacks.extend(getattr(step, "capture_notes", []))
            except Exception as e:  # noqa: BLE001
                err = e
                fallbacks.append(f"{attempt}: {e!r}")
                print(f"[bench] capture ({attempt}) failed on rank {rank}: {e!r}", file=sys.stderr, flush=True)
                if attempt == "as configured" and step._one_graph_exchange_ok():
                    # a failed capture that HOLDS COLLECTIVES ends the attempt in this process: nothing guarantees that the
                    # stream / communicator is reusable (round 3: carrying on aborted the process) -- a pre-flight child exits
                    # non-zero here and its parent takes the staged graphs
                    raise
                torch.cuda.synchronize()
                if args.strict_capture or attempt == "eager":
                    raise
            if all_ranks_ok(err is None):
                break
            if err is None:
                fallbacks.append(f"{attempt}: failed on another rank")
            run = None
        if run is None:
            raise RuntimeError("bench.py: no execution mode worked")
    else:
        eager_step()  # materialise the flat buffers
        run = eager_step

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(warmup):
        run()

    def timed_block():
        """EXACTLY ``steps`` steps between barrier + synchronize on both sides: milliseconds per step of this rank."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        torch.cuda.synchronize()
        barrier()
        return (time.perf_counter() - t0) * 1e3 / steps

    def over_ranks(v, op):
        if world <= 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if coord_cpu else device)
        torch.distributed.all_reduce(t, op=op)
        return float(t.item())

    # The driver's K = 20 steps of the headline workload are a 30 ms window on boxes that differ by +-4 % and processes that
    # land in one of two modes a few percent apart: the K-step block is REPEATED until at least ``min_timed_s`` of stepping
    # has been timed (every block bracketed as above, MAX over ranks per block) and the MEDIAN block is reported; ``timed_s``
    # / ``timed_blocks`` / ``block_ms`` say what was timed.
    blocks = [over_ranks(timed_block(), torch.distributed.ReduceOp.MAX)]
    want = 1
    if min_timed_s > 0 and blocks[0] * steps * 1e-3 < min_timed_s:
        want = min(200, int(min_timed_s / max(blocks[0] * steps * 1e-3, 1e-6)) + 1)
    want = int(over_ranks(float(want), torch.distributed.ReduceOp.MAX))  # (every rank runs the same number of blocks)
    while len(blocks) < want:
        blocks.append(over_ranks(timed_block(), torch.distributed.ReduceOp.MAX))
    srt = sorted(blocks)
    ms = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    # ``steps`` of the JSON line stays K (the contract's block length); ``steps_timed`` = K x blocks is what was really timed
    timing = {"timed_s": sum(blocks) * steps * 1e-3, "timed_blocks": len(blocks), "steps_timed": steps * len(blocks),
              "block_ms_min": srt[0], "block_ms_max": srt[-1]}

    # N ranks: what the gradient exchange costs the step BEYOND what backward hides -- the same steps with the collectives
    # left out of the exchange path (conversion, stream hand-offs, per-chunk Adam all still run: GradSync.skip_collectives)
    exchange = None
    if sync is not None and world > 1:
        recapture = getattr(step, "_graph_has_exchange", False)  # (a graph that holds the collectives is captured again without them)
        # the leg steps every rank on its LOCAL gradients: parameters, moments and the step counter are put back afterwards, so
        # that whatever follows in this process is the replicated run again
        snap = (opt.flat_p.clone(), opt.flat_m.clone(), opt.flat_v.clone(), opt.step_count)
        sync.skip_collectives = True
        try:
            if recapture:
                step.capture(dev, fused_merged, warmup=0)
            for _ in range(max(2, warmup // 2)):
                run()
            dry = over_ranks(timed_block(), torch.distributed.ReduceOp.MAX)
        finally:
            sync.skip_collectives = False
            if recapture:
                step.capture(dev, fused_merged, warmup=0)
        with torch.no_grad():
            opt.flat_p.copy_(snap[0])
            opt.flat_m.copy_(snap[1])
            opt.flat_v.copy_(snap[2])
        opt.step_count = snap[3]  # (the device-side counter follows at the next step: FlatAdam.sync_hyper_source)
        opt.refresh_shadows()
        del snap
        torch.cuda.synchronize()
        exchange = {"ms_per_step_without_collectives": dry, "exposed_ms_per_step": ms - dry,
                    "bytes_per_step": opt.flat_g.numel() * (2 if args.grad_compress == "bf16" else 4),
                    "rccl_ranks": torch.distributed.get_world_size(group), "backend": torch.distributed.get_backend(group)}
    if getattr(args, "stamps", False) and rank == 0:
        prev = 0.0
        for name, us in sorted(ops.stamps_read(), key=lambda kv: kv[1]):
            print(f"[stamp] {us:9.1f} us  (+{us - prev:7.1f})  {name}", file=sys.stderr)
            prev = us

    rl, table = (None, {})
    # (with several ranks EVERY rank takes these eager steps: they issue the gradient collectives, and a rank 0 stepping
    #  alone would wait for its peers for ever -- found by running two real ranks, --one-gpu-gloo; rank 0 reports)
    if (rank == 0 or world > 1) and want_roofline and not args.no_roofline:
        try:
            # per-kernel durations are taken with every launch on ONE stream (heads, aux tasks and weight gradients
            # serialised), the way rocprofv3 --kernel-trace times them: the committed profile must agree with them
            saved = (step.parallel_heads, step.wgrad_side_streams)
            step.parallel_heads = step.wgrad_side_streams = False
            g1 = getattr(step, "graphone", None)
            if g1 is not None:
                saved_g1, g1.parallel_tasks = g1.parallel_tasks, False
            try:
                rl, table = roofline(ops, eager_step, args.compute, args=args)
            finally:
                step.parallel_heads, step.wgrad_side_streams = saved
                if g1 is not None:
                    g1.parallel_tasks = saved_g1
            # ... and once more with the step's OWN stream structure (heads, weight gradients and Adam beside the dX chain, as the
            # captured graph replays them): the kernel with the largest summed time THERE -- co-running launches slow each
            # other down, so this is the figure a trace of the replayed graph shows (round-2 verdict: the grouped
            # weight-gradient launch, invisible in the serialised table)
            try:
                rl2, _ = roofline(ops, eager_step, args.compute, args=args)
                if rl and rl2:
                    # TOP LEVEL = the kernel with the largest summed time in the step's production stream structure (what a trace of
                    # the replayed graph shows: round-4 verdict, measurement repair ii); the serialised table's dominant kernel -- the
                    # figure the committed rocprofv3 --stats summary can be checked against launch by launch -- rides as "serialised"
                    ser = {k: rl[k] for k in ("kernel", "rocprof_symbol", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                              "launches_per_step", "alg_per_launch", "traffic") if k in rl}
                    ser["timing"] = "HIP events with every launch of the eager steps on ONE stream (the way rocprofv3 --kernel-trace times them)"
                    top = {k: rl2[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "rocprof_symbol",
                                               "launches_per_step", "avg_launch_us", "alg_per_launch") if k in rl2}
                    top["timing"] = ("HIP events on each launch's own stream, eager steps issued with the production stream structure "
                                     "(kernels co-run as in the replayed graph); dominant = largest summed time per step")
                    top["step"] = rl["step"]
                    top["serialised"] = ser
                    rl = top
            except Exception as e:  # noqa: BLE001
                rl["replay_error"] = repr(e)
        except Exception as e:  # the headline number must still be printed
            rl = {"error": repr(e)}
    if rl and "step" in rl:
        rl["step"]["frac"] = rl["step"]["lower_bound_ms"] / ms
    # PMC passes are per configuration on ONE GPU (profiles/pmc_<key>.json): none for N ranks or the exchange dry run
    pmc_config = world == 1 and not args.exchange_dry_run and pmc_file(args).exists()
    if rl and "traffic" in rl and not pmc_config:
        rl["traffic"], rl["traffic_source"] = None, f"not measured for this configuration (no profiles/pmc_{pmc_key(args)}.json)"
        if isinstance(rl.get("serialised"), dict) and "traffic" in rl["serialised"]:
            rl["serialised"]["traffic"] = None
    cb = None
    if sds is not None:
        try:
            ego = None
            if args.workload == "egopack_oscc":
                ego = {"primary": "oscc", "others": ("ar", "lta", "pnr"), "k": args.graphone_k, "depth": args.graphone_depth}
            cb = cpu_baseline(sds, names, dev, weights, egopack=ego)
        except Exception as e:
            cb = {"error": repr(e)}
    # replicated run: finite parameters, the same on every rank (checked over the coordination group)
    params_ok = bool(torch.isfinite(opt.flat_p).all().item())
    if world > 1:
        h = opt.flat_p.double().sum().item()
        params_ok = params_ok and over_ranks(h, torch.distributed.ReduceOp.MAX) == over_ranks(h, torch.distributed.ReduceOp.MIN)
    return {"params_ok": params_ok,
            "ms": ms, "roofline": rl, "table": table, "cpu_baseline": cb, "n_params": opt.flat_p.numel(), "capture": capture,
            "fallbacks": fallbacks, "mode": "eager" if capture == "eager" else "graph", "timing": timing, "exchange": exchange}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # asked for N GPUs without a launcher: start the N ranks ourselves, before anything initialises the device here
        sys.exit(spawn_ranks(args.gpus, argv))
    if args.hw_queues:  # opt-in: must be set before anything initialises the device
        os.environ["GPU_MAX_HW_QUEUES"] = str(args.hw_queues)

    lib_path = REPO / "egopack_amd" / "libegopack_hip.so"
    if not lib_path.exists():  # a checkout without the (git-ignored) library: compile it in-tree, once, rank 0 first
        from egopack_amd import build as _build
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            _build.build_library(force=False, verbose=False)
        else:
            while not lib_path.exists():
                time.sleep(1.0)
            time.sleep(2.0)
    from egopack_amd import dist as edist
    # Two groups for N ranks: the DEFAULT group is gloo -- coordination only (mode flags, barriers, max-over-ranks timings;
    # creating it touches no GPU) -- and the gradient exchange runs over an RCCL group created below (``data_group``).
    rank, local_rank, world = edist.init_from_env("gloo")
    if args.one_gpu_gloo:
        local_rank = 0  # every rank on the box's one GPU
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus = {world}", file=sys.stderr)
    if rank != 0:  # only rank 0 reports: nothing else (RCCL's C-level stdout banner included) may reach the shared stdout
        sys.stdout.flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    # which capture the N-rank step takes: resolved BEFORE this process touches the GPU (the pre-flight children are started
    # by a process without a device context)
    rccl_exchange = (world > 1 and not args.one_gpu_gloo) or (world == 1 and args.exchange_dry_run > 1)
    probe = "not run"
    if args.probe_child or not rccl_exchange or args.mode != "graph" or args.staged == "off":
        if args.exchange_graph == "auto":
            args.exchange_graph = "staged"
    elif args.exchange_graph == "auto":
        ok, probe = one_graph_probe(args, argv, rank, world)
        args.exchange_graph = "one" if ok else "staged"
    args.probe_note = probe
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    data_group = None
    if world > 1 and not args.one_gpu_gloo:
        data_group = torch.distributed.new_group(backend="nccl")  # RCCL over xGMI: the gradient exchange

    if args.gemm_knob is not None:
        from egopack_amd import _lib
        for v in str(args.gemm_knob).split(","):
            _lib.load().egk_gemm_set_pipeline(int(v))
    for kv in filter(None, args.egk_tune.split(",")):
        from egopack_amd import _lib
        k, v = kv.split("=")
        if k == "gather_max":  # (development: 0 = the generic max-aggregation kernel, bit-identical)
            _lib.load().egk_gather_max_tune(int(v))
            continue
        _lib.load().egk_tune(int(k), int(v))
    from egopack_amd import ops
    if args.ln_reduce_inline:
        ops._wgrad_ln["side"] = False
    if args.last_wgrad_side:
        ops._last_wgrad["inline"] = False
    if args.csr_split_heavy:
        from egopack_amd import data as _D
        _D.HEAVY_IN_LAUNCH_DEGREE = 0

    res = measure(args, rank, world, device, args.steps, args.warmup, min_timed_s=args.min_timed_s, group=data_group)
    ms = res["ms"]
    if args.probe_child:
        # a pre-flight child of --exchange-graph auto: the one-graph step captured, replayed and (N > 1) re-captured for the
        # exchange-cost leg; exit 0 only if it really ran in that mode and the ranks hold the same finite parameters
        ok = res["capture"] == "one graph incl. the gradient exchange" and bool(res["params_ok"])
        print(f"[bench probe] rank {rank}: capture '{res['capture']}', parameters ok {res['params_ok']}", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(0 if ok else 3)  # (no teardown of the communicator in a process whose only job was to survive)
    seqs_per_step = world * len(WORKLOADS[args.workload][0]) * args.batch

    # reference-precision leg (N = 1, rank 0): the same step in --compute f32 (exact-f32 MFMA, f32 activations / weights --
    # what the reference computes in), a short run, so the record carries a figure at the reference's own precision
    f32_leg = None
    if rank == 0 and world == 1 and args.compute != "f32" and not args.no_f32_leg and args.mode == "graph":
        try:
            import copy
            a32 = copy.copy(args)
            a32.compute = "f32"
            r32 = measure(a32, rank, world, device, args.f32_steps, 3, want_roofline=not args.no_roofline, want_cpu=False)
            f32_leg = {"ms_per_step": r32["ms"], "value": seqs_per_step / (r32["ms"] * 1e-3), "unit": "clip-seqs/s",
                       "steps": args.f32_steps, "capture": r32["capture"], **r32["timing"],
                       "roofline": None if not r32["roofline"] or "error" in r32["roofline"] else
                       {k: r32["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_us",
                                                         "launches_per_step", "step") if k in r32["roofline"]}}
        except Exception as e:  # noqa: BLE001
            f32_leg = {"error": repr(e)}
        finally:
            ops.set_compute(args.compute)

    if rank == 0:
        rl, cb = res["roofline"], res["cpu_baseline"]
        out = {
            "metric": ("clip-seqs/sec training, AR+LTA+PNR multi-task" if args.workload == "mtl"
                       else f"clip-seqs/sec training, {args.workload}"), "value": seqs_per_step / (ms * 1e-3),
            "unit": "clip-seqs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.compute == "f32" else "bf16", "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.workload][1]}: per GPU B={args.batch} seqs/task x T={args.T} nodes, "
                                   f"3x1536-d Omnivore-shaped features, H={args.hidden}, TRN hidden {args.trn_hidden} "
                                   f"(dropout {args.dropout}), depth 3, k=1, Adam; {res['mode']} mode, "
                                   f"{'fused' if not args.no_fused_backbone else 'per-task'} backbone pass",
                       "global_batch": seqs_per_step, "nodes_per_step": seqs_per_step * args.T,
                       "parallelism": f"dp{world}", "trainable_params": res["n_params"],
                       "grad_allreduce": (("bf16" if args.grad_compress == "bf16" else "f32") if world > 1 else None),
                       "capture": res["capture"], "capture_fallbacks": res["fallbacks"],
                       "exchange_graph": ({"mode": args.exchange_graph, "probe": args.probe_note, "ranks_identical": res["params_ok"]}
                                          if (world > 1 or args.exchange_dry_run > 1) else None),
                       "transport": ("gloo, all ranks on ONE GPU (development run of the N-rank path: not a multi-GPU figure)"
                                     if args.one_gpu_gloo else ("RCCL" if world > 1 else None)),
                       "input": (f"assembled in the step by egk_gather_rows from a resident {args.feature_store}-row feature store"
                                 if args.feature_store else "resident in HBM before the timed region"),
                       "master_weights": "f32", "mode": args.compute,
                       "activations": "bf16" if args.compute == "bf16" else "f32"},
            "roofline": rl, "cpu_baseline": cb, "f32": f32_leg,
            **res["timing"],
        }
        if res["exchange"] is not None:  # N ranks: what the driver's SCALE line needs to be read (round-2 verdict #5c)
            out["exchange"] = res["exchange"]
        if args.kernel_table and res["table"]:
            print(json.dumps(res["table"], indent=1), file=sys.stderr)
    else:
        out = None
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if out is not None:
        # RCCL writes a version banner to the C stdout stream (block-buffered when piped): push it out BEFORE the
        # result so that the JSON line is the last line of rank 0's stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
