#!/usr/bin/env python3
"""EgoPack novel-task training (entry point of the reference's main_egopack.py).

    python main_egopack.py enable_graphone=True resume_from=checkpoints/MTL_ar-lta-pnr/checkpoint.pth \
        enabled_tasks=[oscc] graphone.k=4 graphone.depth=3 graphone.residual=True

Loads the multi-task checkpoint (task heads with strict=False: the auxiliary classifiers are new),
builds the frozen per-task prototype banks with ``build_graphone`` over the AR training set
(batch 256, shuffle False, drop_last True), wraps them in ``GraphONE`` and trains the novel task with
late fusion.  ``resume_from`` is a local checkpoint path; as in the reference its STRING also selects
which tasks feed the prototype banks (substring match on the task names, main_egopack.py:301)."""
from __future__ import annotations

import logging
import time
from pathlib import Path

import torch

from egopack_amd import dist as edist
from egopack_amd import engine, ops, train as T
from egopack_amd.config import instantiate
from egopack_amd.data import build_dataloader, multiloader
from graphone import build_graphone
from models.graphONE.graphONE import GraphONE
from models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask
from utils.meters import build_meter_for_dataset
from validate import validate, validate_lta, validate_pnr

logger = logging.getLogger("main_egopack")
RATE_WARMUP_STEPS = 30  # steps of an epoch left out of its logged steady-state rate (eager steps, the capture, first replays)

AUX_ORDER = {"ar": ("lta", "oscc", "pnr"), "oscc": ("ar", "lta", "pnr"), "lta": ("ar", "oscc", "pnr"),
             "pnr": ("ar", "lta", "oscc")}  # other_tasks lists of the reference's validation calls (main_egopack.py:377-448)


def validate_metrics(epoch, model, tasks, graphone, weights, dsets_val, loaders, late_fusion=True, validate_all=False,
                     device="cuda"):
    """Task metrics with the GraphONE interaction for the novel task(s) (reference main_egopack.py:374-448)."""
    out = {}
    for t in ("ar", "oscc", "lta", "pnr"):
        if not (validate_all or weights.get(t, 0) > 0):
            continue
        is_egopack = weights.get(t, 0) > 0
        others = [tasks[o] for o in AUX_ORDER[t] if o in graphone.task_labels] if is_egopack else []
        g1 = graphone if is_egopack else None
        meter = build_meter_for_dataset(dsets_val[t], device=device)
        if t == "lta":
            validate_lta(model, loaders[t], meter, tasks[t], others, g1, late_fusion=late_fusion, device=device)
        elif t == "pnr":
            validate_pnr(model, loaders[t], meter, tasks[t], others, g1, late_fusion=late_fusion, device=device)
        else:
            validate(epoch, model, loaders[t], meter, tasks[t], others, g1, late_fusion=late_fusion, device=device)
        meter.all_reduce()  # ranks validated disjoint batches of the split: every rank ends with the totals
        for line in meter.print_logs():
            logger.info("[val %s] %s", t, line)
        out[t] = {k: v for k, v in meter.get_logs().items() if isinstance(v, (int, float))}
    return out


def train(epoch, step: engine.EgoPackStep, loaders, weights, device="cuda", store=None):
    """One epoch (reference main_egopack.train :64-159)."""
    order = ("ar", "lta", "oscc", "pnr")
    it, seqs, mark = 0, 0, None
    hosts = (dict(zip(order, batch)) for batch in multiloader([loaders[t] for t in order], [weights[t] for t in order]))
    # batch i + 1 is collated and copied to the device (copy stream) while step i runs; per-task backbone passes here
    for batches, _ in engine.StagedBatches(hosts, device, order, fused=False, store=store, dtype=ops.act_dtype(), step=step):
        total, _ = step.train_step(batches)  # eager for the first steps, then the captured step
        seqs += sum(int(b.num_graphs) for b in batches.values() if b is not None)
        it += 1
        if it == RATE_WARMUP_STEPS and torch.cuda.is_available():
            torch.cuda.synchronize()
            mark = (it, time.perf_counter(), seqs)
    if mark is not None and it > mark[0]:
        torch.cuda.synchronize()
        dt = time.perf_counter() - mark[1]
        logger.info("epoch %d: steady state %.3f ms/step, %.0f clip-seqs/s on this rank (%d steps after the first %d); "
                    "device memory %.0f MB allocated, %.0f MB reserved", epoch, dt * 1e3 / (it - mark[0]), (seqs - mark[2]) / dt,
                    it - mark[0], mark[0], torch.cuda.memory_allocated() / 2 ** 20, torch.cuda.memory_reserved() / 2 ** 20)
    logger.info("epoch %d: %d iterations, last objective %.4f", epoch, it, float(total))
    lc = getattr(step, "loop_counts", None)
    if lc is not None:  # how many steps replayed the captured graph and how many ran eagerly (shape changes, warm-up)
        logger.info("epoch %d: %d steps replayed the captured step, %d ran eagerly", epoch, lc["replayed"], lc["eager"])
        step.loop_counts = {"replayed": 0, "eager": 0}
    return it


def main(argv=None):
    cfg = T.load_config(argv)
    if not cfg.enable_graphone:
        logging.warning("Invalid configuration. Aborting!")
        return
    rank, local_rank, world = T.env_ranks()
    T.setup_logging(rank)
    T.cap_host_threads(int(cfg.get("host_threads", 8)))
    T.seed_everything(cfg, rank)
    ops.set_compute(cfg.compute)
    weights = T.task_weights(cfg)

    # (before anything touches the GPU: collation workers are then a plain fork, data.BatchLoader.start_workers)
    dsets_train, dsets_val = T.build_datasets(cfg, "train"), T.build_datasets(cfg, cfg.validation_split)
    dl_train = T.build_loaders(cfg, dsets_train, True, rank, world)
    dl_val = T.build_loaders(cfg, dsets_val, False, rank, world)  # batch-sharded; meters are summed across ranks
    T.start_loader_workers(dl_train)
    rank, local_rank, world = edist.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    H = cfg.model.hidden_size
    model = instantiate(cfg.model, input_size=dsets_train["ar"].features_size,
                        num_segments=cfg.dataset_recognition.num_segments, _recursive_=False).to(device)
    kw = dict(dropout=cfg.task_dropout, head_dropout=cfg.task_head_dropout)
    tasks = {
        "ar": RecognitionTask(H, H, heads=dsets_train["ar"].num_class_labels, aux_tasks=("oscc", "lta", "pnr"), **kw),
        "oscc": OSCCTask(H, H, aux_tasks=("ar", "lta", "pnr"), average_logits=True, **kw),
        "lta": LTATask(H, H, heads=dsets_train["lta"].num_class_labels, aux_tasks=("ar", "oscc", "pnr"), **kw),
        "pnr": PNRTask(H, H, aux_tasks=("ar", "oscc", "lta"), **kw),
    }
    for t in tasks.values():
        t.to(device)
    if cfg.resume_from:
        logger.info("resuming from %s", cfg.resume_from)
        T.load_checkpoint(cfg.resume_from, model, tasks, strict_tasks=False, device=device)

    # datasets that index a device-resident feature table (dataset_*=synthetic_resident; the reference's .npy files loaded
    # into HBM): the training step gathers its rows on the device, the evaluation and prototype-bank passes through an adapter
    store, store_val = T.build_feature_store(dsets_train, device), T.build_feature_store(dsets_val, device)
    if store_val is not None:
        dl_val = {t: T.ResidentLoader(l, store_val, device, ops.act_dtype()) for t, l in dl_val.items()}
    bank_tasks = [tasks[t] for t in ("ar", "oscc", "lta", "pnr") if tasks[t].name in str(cfg.resume_from)]
    bank_loader = build_dataloader(dsets_train["ar"], 256, False, cfg.num_workers, True, cfg.seed, rank=rank, world_size=world,
                                   shard="batches")
    if store is not None:
        bank_loader = T.ResidentLoader(bank_loader, store, device, ops.act_dtype())
    banks = build_graphone(model, tasks["ar"], bank_tasks, dataloader=bank_loader,
                           device=device)  # partial fp64 banks are summed across ranks inside
    graphone = GraphONE(banks, **cfg.graphone).to(device)

    wd = cfg.optimizer.weight_decay
    params = [*model.configure_optimizers(wd), *(p for t in ("ar", "oscc", "lta", "pnr") for p in tasks[t].configure_optimizers(wd)),
              *graphone.parameters()]
    if world > 1:
        for p in params:  # same start everywhere (seeded identically; broadcast makes it unconditional)
            torch.distributed.broadcast(p.data, src=0)
    optimizer = T.build_optimizer(cfg, params)
    scheduler = T.build_scheduler(cfg, optimizer)
    sync = edist.GradSync(world) if world > 1 else None
    step = engine.EgoPackStep(model, tasks, graphone, weights, optimizer,
                              backprop_temporal_graph=cfg.backprop_temporal_graph,
                              temporal_graph_train_mode=cfg.temporal_graph_train_mode, sync=sync)
    step.use_graph = bool(cfg.get("use_graph", True))
    step.exact_graph_ln = bool(cfg.get("exact_graph_ln", False))
    for epoch in range(1, cfg.num_epochs + 1):
        train(epoch, step, dl_train, weights, device, store=store)
        scheduler.step()
        validate_metrics(epoch, model, tasks, graphone, weights, dsets_val, dl_val, late_fusion=cfg.late_fusion,
                         validate_all=cfg.validate_all_tasks, device=device)  # all ranks: the split is sharded by batch
    if cfg.save_model and step.sync is not None:
        step.sync.gather_moments(optimizer)  # (sharded update: a collective, every rank; a no-op otherwise)
    if cfg.save_model and rank == 0:
        name = f"{cfg.artifact_prefix}_egopack_" + "-".join(sorted(t for t, w in weights.items() if w > 0))
        T.save_checkpoint(Path(cfg.checkpoint_dir) / name / "checkpoint.pth", model, tasks, cfg.num_epochs,
                          graphone=graphone, optimizer=optimizer)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
