#!/usr/bin/env python3
"""Multi-task pre-training of the temporal backbone (entry point of the reference's main_temporal.py).

    python main_temporal.py k=1 batch_size=16 num_epochs=40 model.temporal_pooling.hidden_size=1024 \
        enabled_tasks=[ar,lta,pnr] save_model=True
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 main_temporal.py ...      # data parallel

Same configuration keys and training semantics as the reference (zero_grad -> backbone forward for
every enabled task batch -> head -> weight * loss.mean() summed -> backward -> Adam; cosine schedule
stepped per epoch; ``multiloader`` restarts exhausted loaders).  In the last epochs every enabled task is
validated with its meter (``validate`` / ``validate_lta`` / ``validate_pnr`` + ``utils.meters``, reference
main_temporal.py:340-400); W&B logging is not reproduced, checkpoints are written locally with the reference's key
layout."""
from __future__ import annotations

import logging
import time
from pathlib import Path

import torch

from egopack_amd import dist as edist
from egopack_amd import engine, ops, train as T
from egopack_amd.config import instantiate
from egopack_amd.data import multiloader
from models.tasks import LTATask, OSCCTask, PNRTask, RecognitionTask
from utils.meters import build_meter_for_dataset
from validate import validate, validate_lta, validate_pnr

logger = logging.getLogger("main_temporal")
RATE_WARMUP_STEPS = 30  # steps of an epoch left out of its logged steady-state rate (eager steps, the capture, first replays)


def train(epoch, step: engine.MTLStep, loaders, weights, device="cuda", store=None):
    """One epoch (reference main_temporal.train :49-134)."""
    step.model.train()
    for t in step.tasks.values():
        t.train()
    order = ("ar", "lta", "oscc", "pnr")
    it = 0
    step.loss_sums()  # (clears the per-task loss sums: they accumulate inside the step, on the device, until the epoch ends)
    hosts = (dict(zip(order, batch)) for batch in multiloader([loaders[t] for t in order], [weights[t] for t in order]))
    # batch i + 1 is built and copied to the device (staging thread, copy stream) while step i runs
    mark = None  # (iteration, wall clock, sequences so far) once the eager steps and the capture are behind: steady-state rate
    seqs = 0
    for batches, merged in engine.StagedBatches(hosts, device, order, fused=step.fused, store=store, dtype=ops.act_dtype(), step=step):
        step.train_step(batches, merged)  # eager for the first steps, then the captured step (no launch of this loop between two)
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
    logger.info("epoch %d: %d iterations, train loss %s", epoch, it,
                {t: round(s_ / max(n_, 1), 4) for t, (s_, n_) in step.loss_sums().items() if n_})
    lc = getattr(step, "loop_counts", None)
    if lc is not None:  # how many steps replayed the captured graph and how many ran eagerly (shape changes, warm-up)
        logger.info("epoch %d: %d steps replayed the captured step, %d ran eagerly", epoch, lc["replayed"], lc["eager"])
        step.loop_counts = {"replayed": 0, "eager": 0}
    return it


@torch.no_grad()
def validate_losses(step: engine.MTLStep, loaders, device="cuda"):
    step.model.eval()
    for t in step.tasks.values():
        t.eval()
    out = {}
    for t in step.enabled:
        s, n = 0.0, 0
        for b in loaders[t]:
            _, vectors, _ = step.losses({t: b.to(device)})
            s += float(vectors[t].sum())
            n += vectors[t].numel()
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:  # batch-sharded split
            tot = torch.tensor([s, n], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tot)
            s, n = float(tot[0]), int(tot[1])
        out[t] = s / max(n, 1)
    return out


def validate_metrics(epoch, model, tasks, enabled, dsets_val, loaders, device="cuda"):
    """Task metrics of every enabled task (reference main_temporal.py:340-400)."""
    out = {}
    for t in enabled:
        meter = build_meter_for_dataset(dsets_val[t], device=device)
        if t == "lta":
            validate_lta(model, loaders[t], meter, tasks[t], device=device)
        elif t == "pnr":
            validate_pnr(model, loaders[t], meter, tasks[t], device=device)
        else:
            validate(epoch, model, loaders[t], meter, tasks[t], device=device)
        meter.all_reduce()  # ranks validated disjoint batches of the split: every rank ends with the totals
        for line in meter.print_logs():
            logger.info("[val %s] %s", t, line)
        out[t] = {k: v for k, v in meter.get_logs().items() if isinstance(v, (int, float))}
    return out


def main(argv=None):
    cfg = T.load_config(argv)
    rank, local_rank, world = T.env_ranks()
    T.setup_logging(rank)
    T.cap_host_threads(int(cfg.get("host_threads", 8)))
    T.seed_everything(cfg, rank)
    ops.set_compute(cfg.compute)
    weights = T.task_weights(cfg)
    logger.info("task weights: %s", weights)
    artifact = f"{cfg.artifact_prefix}_" + "-".join(sorted(t for t, w in weights.items() if w > 0))

    # datasets, loaders and their collation processes come first: nothing has touched the GPU yet, so the workers are a
    # plain fork of a process without HIP state (data.BatchLoader.start_workers)
    dsets_train, dsets_val = T.build_datasets(cfg, "train"), T.build_datasets(cfg, cfg.validation_split)
    assert len({d.features_size for d in dsets_train.values()}) == 1, "all tasks must share the input feature size"
    dl_train = T.build_loaders(cfg, dsets_train, True, rank, world)
    dl_val = T.build_loaders(cfg, dsets_val, False, rank, world)  # batch-sharded; meters are summed across ranks
    T.start_loader_workers(dl_train)
    rank, local_rank, world = edist.init_from_env()
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # datasets that index a device-resident feature table (dataset_*=synthetic_resident; the reference's .npy files
    # loaded into HBM): one table per split, the training step gathers its rows on the device, evaluation batches get
    # theirs through an adapter
    store, store_val = T.build_feature_store(dsets_train, device), T.build_feature_store(dsets_val, device)
    if store_val is not None:
        dl_val = {t: T.ResidentLoader(l, store_val, device, ops.act_dtype()) for t, l in dl_val.items()}

    H = cfg.model.hidden_size
    model = instantiate(cfg.model, input_size=dsets_train["ar"].features_size,
                        num_segments=cfg.dataset_recognition.num_segments, _recursive_=False).to(device)
    tasks = {
        "ar": RecognitionTask(H, H, heads=dsets_train["ar"].num_class_labels, dropout=cfg.task_dropout, head_dropout=cfg.task_head_dropout),
        "oscc": OSCCTask(H, cfg.oscc_feat_size, dropout=cfg.task_dropout, head_dropout=cfg.task_head_dropout, loss_func=cfg.oscc_loss),
        "lta": LTATask(H, H, heads=dsets_train["lta"].num_class_labels, dropout=cfg.task_dropout, head_dropout=cfg.task_head_dropout),
        "pnr": PNRTask(H, H, dropout=cfg.task_dropout, head_dropout=cfg.task_head_dropout),
    }
    for t in tasks.values():
        t.to(device)
    wd = cfg.optimizer.weight_decay
    params = [*model.configure_optimizers(wd), *(p for t in ("ar", "oscc", "lta", "pnr") for p in tasks[t].configure_optimizers(wd))]
    if world > 1:
        for p in params:  # same start everywhere (seeded identically; broadcast makes it unconditional)
            torch.distributed.broadcast(p.data, src=0)
    optimizer = T.build_optimizer(cfg, params)
    scheduler = T.build_scheduler(cfg, optimizer)
    compress = str(cfg.get("grad_compress", "none"))  # element type of the gradient all-reduce: none (f32) | bf16
    sync = edist.GradSync(world, compress=compress) if world > 1 else None
    dry = int(cfg.get("exchange_dry_run", 0))
    if world == 1 and dry > 1:
        # development / test knob: run the N-rank exchange path (conversion, RCCL all-reduce on a 1-rank group, 1/N gradient
        # scale, staged backward) on ONE GPU -- everything of the N-GPU step except the peers' contributions
        edist.init_single_rank_group()  # (no-op when the caller already owns a group; free rendezvous port otherwise)
        sync = edist.GradSync(dry, compress=compress)
    step = engine.MTLStep(model, tasks, T.build_criteria(dsets_train), weights, optimizer,
                          fused_backbone=cfg.fused_backbone, sync=sync)
    step.use_graph = bool(cfg.get("use_graph", True))
    step.exact_graph_ln = bool(cfg.get("exact_graph_ln", False))  # several ranks: graph-LN statistics over the GLOBAL batch

    first_epoch = 1
    ckpt_path = Path(cfg.checkpoint_dir) / artifact / "checkpoint.pth"
    if cfg.resume_from:  # continue an interrupted run: weights, Adam moments, schedule, epoch counter
        ck = T.load_checkpoint(cfg.resume_from, model, tasks, strict_tasks=True, device=device, optimizer=optimizer,
                               scheduler=scheduler, loaders=dl_train)
        first_epoch = int(ck.get("epoch", 0)) + 1
        logger.info("resumed from %s at epoch %d", cfg.resume_from, first_epoch)
    metrics = None
    for epoch in range(first_epoch, cfg.num_epochs + 1):
        train(epoch, step, dl_train, weights, device, store=store)
        scheduler.step()
        logger.info("learning rate -> %.6g", scheduler.get_last_lr()[0])
        if cfg.save_model and cfg.get("save_every", 0) and epoch % cfg.save_every == 0 and sync is not None:
            sync.gather_moments(optimizer)  # (sharded update: a collective, every rank; a no-op otherwise)
        if cfg.save_model and cfg.get("save_every", 0) and epoch % cfg.save_every == 0 and rank == 0:
            T.save_checkpoint(ckpt_path, model, tasks, epoch, optimizer=optimizer, scheduler=scheduler, loaders=dl_train)
        if epoch >= cfg.num_epochs - 5:  # all ranks: the validation split is sharded by batch
            logger.info("validation losses: %s", validate_losses(step, dl_val, device))
            metrics = validate_metrics(epoch, model, tasks, step.enabled, dsets_val, dl_val, device)
    if cfg.num_epochs < first_epoch and cfg.get("validate_untrained", False):  # (num_epochs=0: metrics of the initial state)
        metrics = validate_metrics(0, model, tasks, step.enabled, dsets_val, dl_val, device)
    if cfg.save_model and sync is not None:
        sync.gather_moments(optimizer)
    if cfg.save_model and rank == 0:
        T.save_checkpoint(ckpt_path, model, tasks, cfg.num_epochs, optimizer=optimizer, scheduler=scheduler, loaders=dl_train)
    if world > 1:
        torch.distributed.destroy_process_group()
    # (callers that drive main() from Python -- the tests -- get the last validation metrics and the trained modules)
    return {"metrics": metrics, "model": model, "tasks": tasks, "step": step, "loaders": dl_train, "val_datasets": dsets_val}


if __name__ == "__main__":
    main()
