"""Validation meters: the core metrics of reference utils/meters/ego4d.py without torchmetrics,
editdistance, W&B tables or t-SNE plots.

Every ``update`` works on the logits / labels where they are (HBM): one ``egk_label_rank`` launch per head gives
the rank of the ground-truth class, and all top-k accuracies, per-class recalls and supports are integer counts of
it kept in small device tensors -- nothing is copied to the host before ``compute`` / ``get_logs``.

    reference class (ego4d.py)        here                 keys kept
    Ego4dRecognitionMeter :34-203     RecognitionMeter     verbs/nouns_top{1,2,3,5}, verbs/nouns_mc, *_class_acc,
                                                           *_calibration_erorr (the reference's spelling), *_brier_score, loss
    Ego4dAnticipationMeter :206-289   AnticipationMeter    *_accuracy_top{1,2,3,5}, *_recall_top{1,2,3,5}, loss
    Ego4dOSCCMeter :292-318           OSCCMeter            accuracy, loss
    Ego4dPNRMeter :321-377            PNRMeter             accuracy, recall, auroc, localization_error, loss
    Ego4dLTAMeter :380-453            LTAMeter             verbs_ed, nouns_ed (+ verbs/nouns_top1), loss
Dropped: confusion matrices, per-class loss tables, feature dumps (reporting only).

Several ranks (SURVEY §8(e) caveat 5): every meter is a set of SUMS (integer counts, float64 loss / distance sums) plus,
for the PNR AUROC, a list of scores.  ``merge`` adds another meter's state, ``all_reduce`` does the same across the
process group (sum all-reduce of the packed counts, all-gather of the scores), so a validation split sharded batch by
batch (data.BatchLoader ``shard="batches"``) reports exactly the numbers of the single-process pass.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.distributed as dist

from . import _lib, ops

KS = (1, 2, 3, 5)


def label_rank(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """int32 [N]: rank of labels[n] inside logits[n, :] (0 = top-1; ties to the lower class index), -1 where the
    label is negative (ignore_index) -- ``egk_label_rank``."""
    ops._need_gpu(logits, labels)
    if logits.dtype != torch.float32:
        logits = logits.float()
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    labels = labels.to(torch.int64)
    n, c = logits.shape
    rank = torch.empty(n, dtype=torch.int32, device=logits.device)
    ops._ck(_lib.load().egk_label_rank(ops._stream(), ops._p(logits), logits.stride(0), ops._p(labels), labels.stride(0),
                                       ops._p(rank), n, c), "egk_label_rank")
    return rank


def edit_distances(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """int32 [N, K] Levenshtein distances between pred[n, :, k] and label[n, :] -- ``egk_edit_distance``."""
    ops._need_gpu(pred, label)
    pred, label = pred.to(torch.int64), label.to(torch.int64)
    n, z, k = pred.shape
    out = torch.zeros((n, k), dtype=torch.int32, device=pred.device)
    if n == 0 or z == 0 or k == 0:  # empty sequences are at distance 0
        return out
    ops._ck(_lib.load().egk_edit_distance(ops._stream(), ops._p(pred), pred.stride(0), pred.stride(1), pred.stride(2),
                                          ops._p(label), label.stride(0), label.stride(1), ops._p(out), n, z, k),
            "egk_edit_distance")
    return out


class _HeadCounts:
    """hits@k, valid count, per-class hits@k and support of one classification head, on the device."""

    def __init__(self, n_classes: int, device, ks: Sequence[int] = KS):
        self.ks, self.C = tuple(ks), n_classes
        self.hits = torch.zeros(len(self.ks), dtype=torch.int64, device=device)
        self.valid = torch.zeros((), dtype=torch.int64, device=device)
        self.class_hits = torch.zeros((len(self.ks), n_classes), dtype=torch.int64, device=device)
        self.support = torch.zeros(n_classes, dtype=torch.int64, device=device)

    def update(self, logits: torch.Tensor, labels: torch.Tensor):
        rank = label_rank(logits, labels)
        ok = rank >= 0
        lab = torch.where(ok, labels.to(torch.int64), torch.zeros_like(labels, dtype=torch.int64))
        self.valid += ok.sum()
        self.support += torch.bincount(lab, weights=ok.to(torch.float64), minlength=self.C).to(torch.int64)
        for i, k in enumerate(self.ks):
            hit = ok & (rank < k)
            self.hits[i] += hit.sum()
            self.class_hits[i] += torch.bincount(lab, weights=hit.to(torch.float64), minlength=self.C).to(torch.int64)

    def tensors(self) -> List[torch.Tensor]:
        return [self.hits, self.valid, self.class_hits, self.support]

    def accuracy(self, k: int) -> float:  # MulticlassAccuracy(top_k=k, average="micro", ignore_index=-1)
        return float(self.hits[self.ks.index(k)]) / max(int(self.valid), 1)

    def class_accuracy(self, k: int) -> torch.Tensor:  # average=None
        return self.class_hits[self.ks.index(k)].double() / self.support.clamp(min=1).double()

    def mean_class(self, k: int = 1) -> float:
        """average="macro" (= utils/meters/utils.py topk_recall): mean over the classes that occurred."""
        seen = self.support > 0
        return float(self.class_accuracy(k)[seen].mean()) if bool(seen.any()) else 0.0


class _Calibration:
    """torchmetrics MulticlassCalibrationError(num_classes, n_bins, norm, ignore_index=-1) as the reference builds it
    (utils/meters/ego4d.py:52-53, :66-67: 15 bins / l1 = the expected calibration error; 1 bin / l2 = its "Brier score") as
    per-bin SUMS on the device -- count, confidence sum, hit count -- so that it merges across shards like every other meter
    (torchmetrics keeps the confidence of every sample and bins at compute time: the same bins, the same sums)."""

    def __init__(self, n_bins: int, norm: str, device):
        if norm not in ("l1", "l2", "max"):
            raise ValueError(norm)
        self.n_bins, self.norm = n_bins, norm
        self.bounds = torch.linspace(0, 1, n_bins + 1, dtype=torch.float32, device=device)
        self.count = torch.zeros(n_bins + 1, dtype=torch.int64, device=device)  # (+1: a confidence of exactly 1 has its own bin)
        self.hits = torch.zeros(n_bins + 1, dtype=torch.int64, device=device)
        self.conf = torch.zeros(n_bins + 1, dtype=torch.float64, device=device)

    def update(self, logits: torch.Tensor, labels: torch.Tensor):
        p = logits.detach().float()
        labels = labels.to(torch.int64)
        in_unit = ((p >= 0) & (p <= 1)).all()  # (scores that already are probabilities are taken as they are)
        p = torch.where(in_unit, p, p.softmax(1))
        conf, pred = p.max(dim=1)
        ok = labels != -1
        w = ok.to(torch.float64)
        idx = (torch.bucketize(conf, self.bounds, right=True) - 1).clamp_(0, self.n_bins)
        self.count += torch.bincount(idx, weights=w, minlength=self.n_bins + 1).to(torch.int64)
        self.hits += torch.bincount(idx, weights=w * (pred == labels).to(torch.float64), minlength=self.n_bins + 1).to(torch.int64)
        self.conf += torch.bincount(idx, weights=w * conf.double(), minlength=self.n_bins + 1)

    def tensors(self) -> List[torch.Tensor]:
        return [self.count, self.hits, self.conf]

    def compute(self) -> float:
        cnt = self.count.double()
        total = float(cnt.sum())
        if total == 0:
            return 0.0
        acc = torch.nan_to_num(self.hits.double() / cnt)
        conf = torch.nan_to_num(self.conf / cnt)
        share = cnt / total
        if self.norm == "l1":
            return float((acc - conf).abs().mul(share).sum())
        if self.norm == "max":
            return float((acc - conf).abs().max())
        ce = float(((acc - conf) ** 2 * share).sum())
        return ce ** 0.5 if ce > 0 else 0.0


class BaseMeter:
    """utils/meters/base.py: running mean of the per-batch loss + sample counter."""

    def __init__(self, save_features: bool = False, device="cuda") -> None:
        self.device = torch.device(device)
        self.loss_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        self.loss_n = 0
        self.counter = 0

    def update(self, labels, loss, *args, **kwargs) -> None:
        loss = loss.detach().double()
        if bool(torch.isnan(loss).any()):
            raise RuntimeError("Encountered `nan` values in tensor")  # MeanMetric(nan_strategy="error")
        self.loss_sum += loss.sum().to(self.device)
        self.loss_n += loss.numel()
        self.counter += labels.shape[0]

    def loss(self) -> float:
        return float(self.loss_sum) / max(self.loss_n, 1)

    # ---- combining meters (other shards of the same split) ----------------------------------------------------
    _INTS = ("loss_n", "counter")  # python-int counters
    _LISTS = ()  # attributes holding lists of 1-D tensors that concatenate across shards

    def _sums(self) -> List[torch.Tensor]:
        """Device tensors that add across shards (updated in place)."""
        return [self.loss_sum]

    def merge(self, other: "BaseMeter") -> "BaseMeter":
        """Add the state of a meter that saw another shard of the split."""
        for a, b in zip(self._sums(), other._sums()):
            a += b.to(a.device)
        for n in self._INTS:
            setattr(self, n, getattr(self, n) + getattr(other, n))
        for n in self._LISTS:
            getattr(self, n).extend(t.to(self.device) for t in getattr(other, n))
        return self

    def all_reduce(self, group=None) -> "BaseMeter":
        """Combine the meters of all ranks (every rank ends with the totals).  Two collectives for the sums (one int64,
        one float64 buffer) and, for list state, a size exchange plus one padded all-gather per list."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        world = dist.get_world_size(group)
        sums = self._sums()
        ints = torch.tensor([getattr(self, n) for n in self._INTS], dtype=torch.int64, device=self.device)
        for dtype in (torch.int64, torch.float64):
            parts = [t for t in sums if t.dtype == dtype] + ([ints] if dtype == torch.int64 else [])
            if not parts:
                continue
            flat = torch.cat([t.reshape(-1) for t in parts])
            dist.all_reduce(flat, group=group)
            o = 0
            for t in parts:
                t.copy_(flat[o: o + t.numel()].view(t.shape))
                o += t.numel()
        for n, v in zip(self._INTS, ints.tolist()):
            setattr(self, n, v)
        for n in self._LISTS:
            mine = torch.cat(getattr(self, n)) if getattr(self, n) else None
            count = torch.tensor([0 if mine is None else mine.numel()], dtype=torch.int64, device=self.device)
            counts = [torch.zeros_like(count) for _ in range(world)]
            dist.all_gather(counts, count, group=group)
            counts = [int(c) for c in counts]
            if max(counts) == 0:
                continue
            dtype = self._list_dtype(n)
            pad = torch.zeros(max(counts), dtype=dtype, device=self.device)
            if mine is not None:
                pad[: mine.numel()] = mine.to(dtype)
            got = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(got, pad, group=group)
            setattr(self, n, [torch.cat([g[:c] for g, c in zip(got, counts)])])
        return self

    def _list_dtype(self, name: str) -> torch.dtype:
        return torch.float32

    def print_logs(self) -> List[str]:
        return [f"Loss: {self.loss():.4f}"]

    def get_logs(self, *args, **kwargs) -> Dict[str, float]:
        return {"loss": self.loss()}


class _VerbNounMeter(BaseMeter):
    def __init__(self, dataset, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.dataset = dataset
        self.idx_verbs, self.idx_nouns = dataset.label_names.index("verbs"), dataset.label_names.index("nouns")
        self.verb_labels, self.noun_labels = dataset.class_labels[self.idx_verbs], dataset.class_labels[self.idx_nouns]
        self.verbs = _HeadCounts(len(self.verb_labels), self.device)
        self.nouns = _HeadCounts(len(self.noun_labels), self.device)

    def _sums(self):
        return [*super()._sums(), *self.verbs.tensors(), *self.nouns.tensors()]

    @torch.no_grad()
    def _count(self, logits, labels):
        self.verbs.update(logits[self.idx_verbs].detach(), labels[:, self.idx_verbs])
        self.nouns.update(logits[self.idx_nouns].detach(), labels[:, self.idx_nouns])


class RecognitionMeter(_VerbNounMeter):
    def __init__(self, dataset, *args, **kwargs) -> None:
        super().__init__(dataset, *args, **kwargs)
        # ego4d.py:52-53 / :66-67: calibration error (15 bins, l1) and "Brier score" (1 bin, l2) per head
        self.calibration = {h: (_Calibration(15, "l1", self.device), _Calibration(1, "l2", self.device)) for h in ("verbs", "nouns")}

    def _sums(self):
        return [*super()._sums(), *(t for h in ("verbs", "nouns") for c in self.calibration[h] for t in c.tensors())]

    @torch.no_grad()
    def update(self, logits, labels, *args, **kwargs) -> None:
        super().update(labels, *args, **kwargs)
        self._count(logits, labels)
        for h, i in (("verbs", self.idx_verbs), ("nouns", self.idx_nouns)):
            for c in self.calibration[h]:
                c.update(logits[i], labels[:, i])

    def print_logs(self):
        v, n = self.verbs, self.nouns
        return [f"Verbs Top-1: {v.accuracy(1) * 100:.2f}, Top-2: {v.accuracy(2) * 100:.2f}, Top-3: {v.accuracy(3) * 100:.2f}, Top-5: {v.accuracy(5) * 100:.2f}",
                f"Nouns Top-1: {n.accuracy(1) * 100:.2f}, Top-2: {n.accuracy(2) * 100:.2f}, Top-3: {n.accuracy(3) * 100:.2f}, Top-5: {n.accuracy(5) * 100:.2f}",
                f"Verbs Mean class: {v.mean_class() * 100:.2f}", f"Nouns Mean class: {n.mean_class() * 100:.2f}",
                f"Verbs Brier score: {self.calibration['verbs'][1].compute():.4f}",
                f"Nouns Brier score: {self.calibration['nouns'][1].compute():.4f}",
                *super().print_logs()]

    def get_logs(self, *args, **kwargs):
        out = {}
        for name, h in (("verbs", self.verbs), ("nouns", self.nouns)):
            out.update({f"{name}_top{k}": h.accuracy(k) for k in KS})
            out[f"{name}_mc"] = h.mean_class()
            out[f"{name}_class_acc"] = {"top-1": h.class_accuracy(1).cpu(), "top-2": h.class_accuracy(2).cpu(),
                                        "top-5": h.class_accuracy(5).cpu(), "support": h.support.cpu()}
            out[f"{name}_calibration_erorr"] = self.calibration[name][0].compute()  # (sic: the reference's key, ego4d.py:174)
            out[f"{name}_brier_score"] = self.calibration[name][1].compute()
        return {**out, **super().get_logs()}


class AnticipationMeter(_VerbNounMeter):
    @torch.no_grad()
    def update(self, logits, labels, *args, **kwargs) -> None:
        super().update(labels, *args, **kwargs)
        self._count(logits, labels)

    def get_logs(self, *args, **kwargs):
        out = {}
        for name, h in (("verbs", self.verbs), ("nouns", self.nouns)):
            out.update({f"{name}_accuracy_top{k}": h.accuracy(k) for k in KS})
            out.update({f"{name}_recall_top{k}": h.mean_class(k) for k in KS})  # topk_recall_fast: classes that occur
        return {**out, **super().get_logs()}

    def print_logs(self):
        logs = self.get_logs()
        return [", ".join(f"{k}: {v * 100:.2f}" for k, v in logs.items() if k != "loss"), *super().print_logs()]


class OSCCMeter(BaseMeter):
    def __init__(self, dataset=None, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.dataset = dataset
        self.counts = _HeadCounts(2, self.device, ks=(1,))

    def _sums(self):
        return [*super()._sums(), *self.counts.tensors()]

    @torch.no_grad()
    def update(self, logits, labels, *args, **kwargs) -> None:
        super().update(labels, *args, **kwargs)
        self.counts.update(logits.detach(), labels)

    def print_logs(self):
        return [f"Accuracy: {self.counts.accuracy(1) * 100:.2f}", *super().print_logs()]

    def get_logs(self, *args, **kwargs):
        return {"accuracy": self.counts.accuracy(1), **super().get_logs()}


class PNRMeter(BaseMeter):
    """BinaryAccuracy / BinaryRecall at threshold 0.5 on sigmoid(logits), exact AUROC, key-frame localisation error."""

    def __init__(self, dataset=None, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.dataset = dataset
        self.stats = torch.zeros(4, dtype=torch.int64, device=self.device)  # tp, tn, positives, total
        self.probs: List[torch.Tensor] = []
        self.targets: List[torch.Tensor] = []
        self.loc_err_sum = torch.zeros((), dtype=torch.float64, device=self.device)
        self.loc_n = 0

    _INTS = (*BaseMeter._INTS, "loc_n")
    _LISTS = ("probs", "targets")

    def _sums(self):
        return [*super()._sums(), self.stats, self.loc_err_sum]

    def _list_dtype(self, name):
        return torch.float32 if name == "probs" else torch.uint8

    @torch.no_grad()
    def update(self, logits, labels, batch, start_frame, end_frame, pnr_frame, *args, **kwargs) -> None:
        super().update(labels, *args, **kwargs)
        from .models.tasks.oscc import sequence_ptr
        probs = torch.sigmoid(logits.detach().float())
        t = labels.to(torch.bool)
        pred = probs > 0.5
        self.stats += torch.stack([(pred & t).sum(), (~pred & ~t).sum(), t.sum(), torch.tensor(t.numel(), device=t.device)])
        self.probs.append(probs)
        self.targets.append(t)
        # per-sequence arg-max node (first maximum) by the segment-max kernel on the [N, 1] probabilities
        ptr = sequence_ptr(batch)
        n_seg = ptr.numel() - 1
        col = probs.contiguous().view(-1, 1)
        best = torch.empty((n_seg, 1), dtype=torch.float32, device=col.device)
        arg = torch.empty((n_seg, 1), dtype=torch.int32, device=col.device)
        ops._ck(_lib.load().egk_segment_max_fwd(ops._stream(), ops._p(col), ops._p(ptr), ops._p(best), ops._p(arg), n_seg, 1, 0),
                "egk_segment_max_fwd")
        loc = (arg.view(-1) - ptr[:-1]).double()
        sf, ef, pf = (v.to(col.device).double() for v in (start_frame, end_frame, pnr_frame))
        err_sec = ((ef - sf) / 16 * loc - (pf - sf)).abs() / 30
        self.loc_err_sum += err_sec.sum()
        self.loc_n += n_seg

    def auroc(self) -> float:
        """Exact area under the ROC curve (Mann-Whitney U, ties count 1/2) over everything seen."""
        if not self.probs:
            return 0.0
        p, t = torch.cat(self.probs).double(), torch.cat(self.targets).to(torch.bool)
        n_pos, n_neg = int(t.sum()), int((~t).sum())
        if n_pos == 0 or n_neg == 0:
            return 0.0
        vals, inv, counts = torch.unique(p, return_inverse=True, return_counts=True)
        ends = torch.cumsum(counts, 0).double()
        mid_rank = ends - (counts.double() - 1) / 2  # average 1-based rank of each distinct value
        r_pos = mid_rank[inv][t].sum()
        return float((r_pos - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg))

    def get_logs(self, *args, **kwargs):
        tp, tn, pos, tot = (int(v) for v in self.stats)
        return {"accuracy": (tp + tn) / max(tot, 1), "recall": tp / max(pos, 1), "auroc": self.auroc(),
                "localization_error": float(self.loc_err_sum) / max(self.loc_n, 1), **super().get_logs()}

    def print_logs(self):
        l = self.get_logs()
        return [f"accuracy: {l['accuracy']:.4f}", f"recall: {l['recall']:.4f}", f"auroc: {l['auroc']:.4f}",
                f"localization_error: {l['localization_error']:.4f}", *super().print_logs()]


class LTAMeter(_VerbNounMeter):
    """Edit distance @ Z = 20 over K = 5 sampled futures (ego4d.py:410-433: sequences of 22 nodes, the 2 observed
    nodes dropped), plus top-1 accuracy on the labelled nodes."""

    N_NODES, N_SAMPLES, SKIP = 22, 5, 2

    def __init__(self, dataset, *args, **kwargs) -> None:
        super().__init__(dataset, *args, **kwargs)
        self.N_NODES = int(getattr(dataset, "lta_nodes", self.N_NODES))  # synthetic datasets use their own T
        self.ed_sum = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.ed_n = 0

    _INTS = (*BaseMeter._INTS, "ed_n")

    def _sums(self):
        return [*super()._sums(), self.ed_sum]

    def _edit_distance(self, preds: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        p = preds.reshape(-1, self.N_NODES, self.N_SAMPLES)[:, self.SKIP:]
        l = labels.reshape(-1, self.N_NODES)[:, self.SKIP:]
        return edit_distances(p, l).min(dim=1).values.double() / p.shape[1]

    @torch.no_grad()
    def update(self, logits, labels, predictions, *args, **kwargs) -> None:
        super().update(labels, *args, **kwargs)
        self._count(logits, labels)  # rows with negative labels are skipped by the rank kernel (labels >= 0 filter)
        dv = self._edit_distance(predictions[self.idx_verbs], labels[:, self.idx_verbs])
        dn = self._edit_distance(predictions[self.idx_nouns], labels[:, self.idx_nouns])
        self.ed_sum += torch.stack([dv.sum(), dn.sum()])
        self.ed_n += dv.numel()
        self.last_distances = (dv, dn)

    def get_logs(self, *args, **kwargs):
        v, n = (float(x) / max(self.ed_n, 1) for x in self.ed_sum)
        return {"verbs_ed": v, "nouns_ed": n, "verbs_top1": self.verbs.accuracy(1), "nouns_top1": self.nouns.accuracy(1),
                **super().get_logs()}

    def print_logs(self):
        l = self.get_logs()
        return [f"verbs_ed: {l['verbs_ed']:.4f}", f"nouns_ed: {l['nouns_ed']:.4f}", f"verbs_top1: {l['verbs_top1']:.4f}",
                f"nouns_top1: {l['nouns_top1']:.4f}", *super().print_logs()]


def build_meter_for_dataset(dataset, save_features: bool = False, device="cuda") -> BaseMeter:
    """utils/meters/__init__.py:10-22, keyed on the dataset's ``task`` name (synthetic and Ego4D datasets alike)."""
    kind = getattr(dataset, "task", None) or type(dataset).__name__.lower()
    if "pnr" in kind:
        return PNRMeter(dataset, device=device)
    if "oscc" in kind:
        return OSCCMeter(dataset, device=device)
    if "lta" in kind:
        return LTAMeter(dataset, device=device)
    if "anticipation" in kind:
        return AnticipationMeter(dataset, device=device)
    if "ar" in kind or "recognition" in kind:
        return RecognitionMeter(dataset, save_features=save_features, device=device)
    raise NotImplementedError(f"no meter for dataset {type(dataset).__name__}")
