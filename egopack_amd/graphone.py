"""Prototype-bank builder (reference graphone.py:17-63).

Eval-mode, no-grad pass over the AR loader: per batch keep the labelled nodes, project them with
every task head, and accumulate ``bank_t[verb*|nouns|+noun] += feature`` in float64 plus a label
count; the banks are the per-label means over the seen labels.  The fp64 [|V|*|N|, H] banks live on
the device for the whole pass (one allocation per task instead of one 450 MB temporary per
``scatter`` call).  Rows are added by a label-grouped segmented reduction (one wave per label of the
batch, rows of a label summed in fp32 in node order like the reference's per-batch scatter, then added
to the fp64 row): no atomics, the banks are bitwise reproducible.

Several ranks (SURVEY §8(e) caveat 3): give every rank a loader sharded batch by batch
(data.BatchLoader ``shard="batches"``: same batches as the single-process pass, split round-robin);
the fp64 partial banks and the counts are summed across ranks once at the end, every rank ends with
the same banks.  The fp64 sums only change their association, which the final cast to fp32 absorbs."""
from __future__ import annotations

import logging
from typing import Dict, List

import torch
import torch.distributed as dist

from . import ops

logger = logging.getLogger(__name__)


@torch.no_grad()
def accumulate_banks(model, ar_task, tasks: List, dataloader, device="cuda"):
    """The pass itself: ({task: fp64 [|V|*|N|, H] row sums}, int64 [|V|*|N|] label counts) of THIS loader's batches."""
    model.eval()
    for t in tasks:
        t.eval()
    feat_size = ar_task.net[-1].out_features
    n_classes = tuple(c[-1].out_features for c in ar_task.classifiers)
    size = n_classes[0] * n_classes[1]
    banks = {t.name: torch.zeros((size, feat_size), dtype=torch.float64, device=device) for t in tasks}
    count = torch.zeros(size, dtype=torch.int64, device=device)
    for data in dataloader:
        # label = verb * |nouns| + noun for labelled nodes, -1 (skipped) otherwise; the nodes are grouped by label where
        # the labels already are (on the host for loader batches: integer work, bit-exact), once per batch
        y = data.y
        labels = torch.where(y[:, 0] != -1, y[:, 0] * n_classes[1] + y[:, 1], torch.full_like(y[:, 0], -1))
        groups = tuple(g.to(device) for g in ops.label_groups(labels))
        data = data.to(device)
        feat = model(data)
        for t in tasks:
            # the label count is incremented once PER TASK, as in the reference where
            # ``all_labels.append(labels)`` sits inside the task loop (graphone.py:50-52): every bank is the
            # per-label mean divided by len(tasks).  Kept for parity (prototypes are L2-normalised downstream).
            ops.scatter_add_rows_f64(t.forward_features(feat), None, banks[t.name], count, groups=groups)
    return banks, count


def finalise_banks(banks: Dict[str, torch.Tensor], count: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Per-label means over the seen labels, fp32 [K, H] per task (reference graphone.py:55-63)."""
    seen = count > 0
    if not bool(seen.any()):
        raise RuntimeError("build_graphone: the loader produced no labelled node (empty loader? the reference uses "
                           "batch 256 with drop_last=True, so the AR split needs at least 256 samples)")
    cnt = count[seen].to(torch.float64).unsqueeze(1)
    return {name: (bank[seen] / cnt).float() for name, bank in banks.items()}


@torch.no_grad()
def build_graphone(model, ar_task, tasks: List, dataloader, device="cuda", group=None) -> Dict[str, torch.Tensor]:
    logger.info("Building graphONE from tasks: %s", ", ".join(t.name for t in tasks))
    banks, count = accumulate_banks(model, ar_task, tasks, dataloader, device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(count, group=group)
        for bank in banks.values():  # |V|*|N| x H fp64 each (450 MB at Ego4D sizes): one collective per task, once
            dist.all_reduce(bank, group=group)
    out = finalise_banks(banks, count)
    logger.info("prototype banks: %s", ", ".join(f"{k} {tuple(v.shape)}" for k, v in out.items()))
    return out
