"""Shared pieces of the two entry points (main_temporal.py / main_egopack.py): config loading, dataset
and loader construction, checkpoints with the reference's key layout, schedulers, per-epoch loops."""
from __future__ import annotations

import logging
import sys
from pathlib import Path
from typing import Dict, Optional

import torch

from . import data as D
from . import ops
from .config import Cfg, compose, instantiate
from .criterion import BCEWithLogitsNone, CrossEntropyNone, MetricSelectorWrapper
from .optim import FlatAdam

logger = logging.getLogger("egopack")
TASKS = ("ar", "oscc", "lta", "pnr")
CKPT_KEYS = {"ar": "task/recognition", "oscc": "task/oscc", "lta": "task/lta", "pnr": "task/pnr"}
DSET_GROUP = {"ar": "dataset_recognition", "oscc": "dataset_oscc", "lta": "dataset_lta", "pnr": "dataset_pnr"}


def load_config(argv=None, config_dir: Optional[Path] = None) -> Cfg:
    """``python main_*.py key=value group/sub=name ...`` (Hydra override syntax)."""
    argv = sys.argv[1:] if argv is None else argv
    config_dir = config_dir or Path(__file__).resolve().parents[1] / "configs"
    return compose(config_dir, "defaults", [a for a in argv if "=" in a])


def cap_host_threads(n: int = 8) -> int:
    """The training process does no arithmetic on the host: what its intra-op thread pool runs is the memcpy of the
    step's feature block into the page-locked staging buffer.  With the default pool of one thread per core (128 on the
    MI355X host) that copy stalls for ~90 ms every few steps (per-step host time: median 83 ms, mean 60 ms); with 4-16
    threads it takes < 1 ms (3.0 ms per step for staging + replay bookkeeping, no stalls) -- tools/exp/loop_trace.py."""
    prev = torch.get_num_threads()
    if prev > n:
        torch.set_num_threads(n)
    return prev


def seed_everything(cfg, rank: int):
    if cfg.seed > 0:
        import numpy as np
        np.random.seed(cfg.seed)
        torch.manual_seed(cfg.seed)  # identical parameter init on every rank
        # dropout streams differ per rank; ``dropout_seed_offset`` (the noise-floor runs of the metric comparison) is mixed in with a
        # stride no rank count reaches, so offset k on rank r is not rank r + k's stream of the base run
        ops.manual_seed(cfg.seed * 7919 + rank + int(cfg.get("dropout_seed_offset", 0) or 0) * 1000003)


def task_weights(cfg) -> Dict[str, float]:
    return {t: float(cfg[f"weight_{t}"]) if t in cfg.enabled_tasks else 0.0 for t in TASKS}


def build_datasets(cfg, split: str):
    """One dataset per task with the reference's transforms: RadiusGraph(r=k+0.5) for AR / OSCC / PNR,
    LTATemporalConnectivity(r=k+0.5) for LTA (main_temporal.py:168-235)."""
    out = {}
    for t in TASKS:
        tf = D.LTATemporalConnectivity(r=cfg.k + 0.5, loop=False) if t == "lta" else D.RadiusGraph(r=cfg.k + 0.5, loop=False)
        dcfg = dict(cfg[DSET_GROUP[t]])
        if dcfg["_target_"].endswith(("SyntheticTaskDataset", "SyntheticResidentDataset", "LearnableSyntheticDataset")):
            n_val = int(cfg.get("synthetic_val_samples", 0)) or max(cfg.synthetic_samples // 4, 1)
            dcfg.update(length=cfg.synthetic_samples if split == "train" else n_val,
                        seed=cfg.seed + (0 if split == "train" else 10_000), k=cfg.k)
            if dcfg["_target_"].endswith("SyntheticResidentDataset"):
                dcfg["split"] = split
            out[t] = instantiate(dcfg, transform=tf)
        else:
            out[t] = instantiate(dcfg, split=split, transform=tf)
    return out


def build_feature_store(dsets, device):
    """ONE device-resident feature table for the tasks' datasets when they index one (datasets that expose ``videos``:
    uid -> [frames, F] and deliver ``x_idx``): the reference's four datasets read the same Omnivore file per video, so do
    these.  None for datasets that deliver features themselves."""
    from .feature_store import FeatureStore
    have = [ds for ds in dsets.values() if getattr(ds, "videos", None) is not None]
    if not have:
        return None
    if len(have) != len(dsets):
        raise ValueError("either every task dataset indexes the resident feature table or none does")
    ref = have[0].videos
    for ds in have[1:]:
        if ds.videos.keys() != ref.keys() or any(ds.videos[k].shape != ref[k].shape for k in ref):
            raise ValueError("the task datasets must index ONE feature table (same videos)")
    from . import ops
    # the table is stored in the activation type of the compute mode: f32 under compute=f32 (the reference-precision mode
    # must not round its inputs to bf16 before the first contraction), bf16 otherwise
    return FeatureStore(ref, device=device, dtype=ops.act_dtype())


def resident_batches(loader, store, device, dtype=None):
    """Evaluation-side adapter: the loader's batches on the device with their features gathered from the store."""
    from .feature_store import materialise_features
    for b in loader:
        yield materialise_features(b.to(device), store, dtype)


class ResidentLoader:
    """A loader of index-only batches (``x_idx``) seen as a loader of device batches with features: every batch is moved to
    the device and its rows are gathered from the resident table (evaluation passes, the prototype-bank pass)."""

    def __init__(self, loader, store, device, dtype=None):
        self.loader, self.store, self.device, self.dtype = loader, store, device, dtype

    def __iter__(self):
        return resident_batches(self.loader, self.store, self.device, self.dtype)

    def __len__(self):
        return len(self.loader)


def build_loaders(cfg, dsets, train: bool, rank: int, world: int, batch_size: Optional[int] = None):
    """Training loaders shard by sample (equal steps per rank); evaluation loaders shard by batch, so that every batch
    -- and with it every graph-LayerNorm statistic -- is the one the single-process pass sees."""
    bs = batch_size or cfg.batch_size
    return {t: D.build_dataloader(ds, bs, train, cfg.num_workers, train, seed=cfg.seed, rank=rank, world_size=world,
                                  shard="samples" if train else "batches",
                                  workers=int(cfg.get("loader_workers", 0)) if train else 0)
            for t, ds in dsets.items()}


def env_ranks() -> tuple:
    """(rank, local_rank, world) of the launcher environment, without initialising anything."""
    import os
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def start_loader_workers(loaders) -> None:
    """Create the collation processes of every loader that asks for some (``loader_workers`` > 0).  The entry points call
    this BEFORE anything touches the GPU: the pool is then a plain fork of a process without HIP state (no runtime locks,
    threads or pinned mappings inherited, the dataset shared copy-on-write instead of pickled to every worker)."""
    for dl in loaders.values():
        if getattr(dl, "workers", 0) > 0:
            dl.start_workers()


def build_criteria(dsets):
    return {"ar": MetricSelectorWrapper(CrossEntropyNone(), dsets["ar"]),
            "lta": MetricSelectorWrapper(CrossEntropyNone(), dsets["lta"]),
            "oscc": CrossEntropyNone(), "pnr": BCEWithLogitsNone()}


def build_optimizer(cfg, params):
    """``_target_: torch.optim.Adam`` of the config is served by the flat-buffer Adam (same arithmetic)."""
    ocfg = dict(cfg.optimizer)
    target = ocfg.pop("_target_")
    if target != "torch.optim.Adam":
        raise ValueError(f"optimizer {target}: only torch.optim.Adam is on the hot path")
    return FlatAdam(params, **ocfg)


def build_scheduler(cfg, optimizer):
    sched = instantiate(cfg.lr_scheduler, optimizer=optimizer)
    if cfg.use_warmup:
        sched = torch.optim.lr_scheduler.ChainedScheduler(
            [torch.optim.lr_scheduler.LinearLR(optimizer, 0.001, 1, 5), sched])
    return sched


def save_checkpoint(path: Path, model, tasks, epoch: int, graphone=None, optimizer=None, scheduler=None, loaders=None):
    """Reference key layout (main_temporal.py:410-417, main_egopack.py:453-460) + what the reference does not keep and
    a resumed run needs: the optimiser state (torch.optim.Adam's per-parameter layout) and the schedule state."""
    path.parent.mkdir(parents=True, exist_ok=True)
    ckpt = {"temporal_graph": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "epoch": epoch}
    for t, key in CKPT_KEYS.items():
        ckpt[key] = {k: v.detach().cpu().clone() for k, v in tasks[t].state_dict().items()}
    if graphone is not None:
        ckpt["graphone"] = {k: v.detach().cpu().clone() for k, v in graphone.state_dict().items()}
    if optimizer is not None:
        sd = optimizer.state_dict()
        sd["state"] = {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in sd["state"].items()}
        ckpt["optimizer"] = sd
    if scheduler is not None:
        ckpt["scheduler"] = scheduler.state_dict()
    if loaders is not None:  # shuffle generators of the training loaders + dropout streams: exact continuation
        ckpt["rng"] = {"loaders": {t: dl.state_dict() for t, dl in loaders.items() if hasattr(dl, "state_dict")},
                       "dropout": ops.get_rng_state(), "torch": torch.get_rng_state()}
    tmp = path.with_suffix(path.suffix + ".tmp")
    torch.save(ckpt, tmp)
    tmp.replace(path)  # a killed run never leaves a half-written checkpoint behind
    logger.info("saved %s", path)


def load_checkpoint(path, model, tasks, strict_tasks: bool = True, device="cpu", graphone=None, optimizer=None,
                    scheduler=None, loaders=None):
    """Weights (reference layout; a reference checkpoint loads as it is) and, when given and present, GraphONE,
    optimiser and schedule state.  Returns the checkpoint dict (``["epoch"]`` = last finished epoch)."""
    ckpt = torch.load(path, map_location=device, weights_only=False)
    model.load_state_dict(ckpt["temporal_graph"])
    for t, key in CKPT_KEYS.items():
        if ckpt.get(key) is not None:
            tasks[t].load_state_dict(ckpt[key], strict=strict_tasks)
    if graphone is not None and ckpt.get("graphone") is not None:
        graphone.load_state_dict(ckpt["graphone"])
    if optimizer is not None and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
        if getattr(optimizer, "materialised", False):
            optimizer.refresh_shadows()  # the bf16 operand copies follow the f32 parameters just loaded
    if scheduler is not None and ckpt.get("scheduler") is not None:
        scheduler.load_state_dict(ckpt["scheduler"])
    if loaders is not None and ckpt.get("rng") is not None:
        for t, st in ckpt["rng"]["loaders"].items():
            if t in loaders and hasattr(loaders[t], "load_state_dict"):
                loaders[t].load_state_dict(st)
        ops.set_rng_state(ckpt["rng"]["dropout"])
        torch.set_rng_state(ckpt["rng"]["torch"].cpu())
    return ckpt


def setup_logging(rank: int):
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING,
                        format="[%(asctime)s][%(name)s][%(levelname)s] %(message)s")
