"""Every development switch of the package in ONE place: name -> (default, one-line meaning).

A switch is a boolean with a default; the environment overrides it by EXACT name (comma lists):

    EGK_DISABLE=grad_store,compact_heads      turns the named switches off
    EGK_ENABLE=sharded_update                 turns the named switches on   (``name=value`` carries a value: ``value()``)
    EGK_DBG=window_cand                       debug prints / measurement-only orderings (``debug()``)

The code asks ``switches.enabled("name")``; a name that is not registered here is a programming error (KeyError), and a name
in the environment that is not registered is reported once (``check_env``) instead of being silently ignored -- the former
``"name" in os.environ.get("EGK_DISABLE", "")`` tests matched SUBSTRINGS (``oscc_one_pass`` also switched ``one_pass`` off).
Default-off entries are the N-rank modes that a caller (or bench.py's probe) opts into.  Numeric knobs with their own variable are listed in ``KNOBS``.
``python -m egopack_amd.switches`` prints the table.
"""
from __future__ import annotations

import os
import warnings

# name: (default, meaning)
REGISTRY = {
    # ---- the training step's structure (engine.py) --------------------------------------------------------------------------
    "wgrad_grouping": (True, "H x H weight gradients parked and issued six at a time as one grouped launch"),
    "deferred_forks": (True, "forked launches are issued one launch late so that the dX chain keeps its hardware queue under capture"),
    "fork_order": (True, "the mechanism behind deferred_forks (ops.defer_after_next_launch); off: forks are issued at once"),
    "grouped_heads": (True, "the projection heads of the task batches as one chain of grouped launches"),
    "grouped_classifiers": (True, "the classifier banks of AR and LTA as one chain of grouped launches"),
    "compact_heads": (True, "AR / LTA heads run on the labelled rows only"),
    "fused_loss": (True, "the cross entropy emits its gradient in the same launch (the seed of backward is known)"),
    "ce_multi": (True, "the cross entropies of the banked tasks as one launch"),
    "rowdot_head": (True, "one-logit / two-logit heads as one row pass (classifier + loss + gradients)"),
    "oscc_one_pass": (True, "the OSCC head (max pool, 2-logit classifier, cross entropy) as pool + one launch"),
    "objective_rider": (True, "the reported objective rides with the parked weight gradients' next flush"),
    "heads_flush": (True, "the heads' parked weight gradients go out beside the first links of the backbone's dX chain"),
    "ln_fusion": (True, "graph-LayerNorm statistics are taken in the epilogue of the contraction that produces its input"),
    "tail_group": (True, "the step ends with ONE grouped launch of the temporal pooling's weight gradients"),
    "early_adam": (True, "EgoPack step: Adam over everything but the pooling's slots beside the last weight-gradient launch"),
    "graphone_adam": (True, "EgoPack step: Adam over GraphONE's slice beside the backbone's backward"),
    "exchange_early_adam": (True, "one-graph exchange: the Adam slices of the regions exchanged so far beside the last weight-gradient launch"),
    "wgrad_handoff": (True, "one-graph exchange: the communication stream (not the backward stream) waits for a region's weight gradients"),
    "grad_store": (True, "gradient slots with one writer per step are stored, not cleared and accumulated (every capture)"),
    "zero_stream": (True, "the gradient buffer is cleared beside the forward pass on a stream of its own"),
    "zero_deferred": (True, "... issued behind the forward pass's first launch"),
    "staging_thread": (True, "training loops: the next steps' batches are built and copied by a thread of their own, two steps ahead"),
    "double_buffered_inputs": (True, "training loops: the step is captured twice, on alternating input buffers filled beside the running replay"),
    "step_gathers_inputs": (True, "training loops: the replayed step gathers the store rows straight into its idle input buffers (no staged feature block)"),
    "staging_priority": (True, "training loops: the staging copy stream and the input-buffer stream are priority streams (hardware queues of their own)"),
    "hyper_in_graph": (True, "the step's Adam constants are computed inside the captured graph"),
    "rng_in_graph": (True, "the dropout offset word moves on inside the captured graph"),
    "rng_early": (True, "... inside the first optimizer launch beside the last weight gradient"),
    "adam_lo": (True, "the Adam launch also writes the low bf16 halves of the three-product weight operands"),
    # ---- EgoPack's precise pass / prototype search ------------------------------------------------------------------------------
    "precise_search": (True, "bf16 modes: the features behind the nearest-prototype search come from a forward-only 'bf16x3' pass"),
    "precise_stream": (True, "the precise pass runs on its own stream beside the training pass's forward"),
    "one_pass": (True, "ONE backbone pass: the bf16 training graph is built from the precise pass's taped results"),
    "primary_early": (True, "the primary projection is issued before the join with the precise pass's stream"),
    "search_ahead": (True, "the prototype searches start when the precise pass ends, on its stream"),
    "grouped_aux": (True, "the auxiliary projections of a batch as grouped launches"),
    "x3_grouped_aux": (True, "... in the three-product mode"),
    "x3_tee": (True, "row kernels of the precise pass also store the bf16 halves of their f32 result (split tee)"),
    "x3_stats_split": (True, "three-product contractions that the policy would split skip the statistics epilogue"),
    "x3_lazy_input": (True, "the bf16 input of the precise pass is not widened in memory"),
    "slab_defer": (True, "the precise pass's split contractions leave their two K slabs to the row kernel that reads the result"),
    "group_ln_tee": (True, "the auxiliary projections' grouped LayerNorm stores the bf16 halves of its result itself"),
    "search_prep": (True, "row norms + bf16 rounding + half rounding of the searched rows as one pass"),
    "grouped_search": (True, "the searches of all auxiliary tasks as one chain of grouped launches"),
    "graphone_grouped": (True, "GraphONE's stages of all auxiliary tasks as one chain of grouped launches"),
    "window_search": (True, "the search as one bf16 product + a proven error window + exact re-rank"),
    "window_f16": (True, "... with the screen on the f16 matrix instructions (narrower window)"),
    "segmax_multi": (True, "the OSCC head's max pools as one launch each way"),
    "ce2_cols_ride": (True, "the OSCC head's classifier gradients ride with the parked weight gradients"),
    "banks_ride": (True, "the classifier banks' weight gradients ride with the parked ones"),
    "proj_park": (True, "the grouped projection's weight gradients are parked"),
    # ---- kernels' host side (ops.py, models) ------------------------------------------------------------------------------------
    "banded_gather": (True, "banded rows of the mean gather take their neighbours from a one-byte code instead of the CSR arrays"),
    "pe_table": (True, "the positional encoding comes from a table of the batch's positions"),
    "f32_wgrad_groups": (True, "exact-f32 weight gradients are grouped like the bf16 ones"),
    "wg4": (True, "four-wave workgroups for the grouped projection's first stage"),
    # ---- several ranks ----------------------------------------------------------------------------------------------------------
    "one_graph_exchange": (False, "N ranks: ONE hipGraph incl. the RCCL collectives (bench.py's probe decides; the attribute also sets it)"),
    "sharded_update": (False, "N ranks: reduce-scatter -> Adam on 1 / world of the buffers -> all-gather instead of all-reduce + full Adam"),
    "adam_behind_collective": (False, "N ranks: every chunk's Adam slice right behind its collective on the communication stream"),
}

DEBUG = {
    "window_cand": "print the candidate statistics of eager window searches",
    "serial_precise": "measurement: the precise pass and the training pass one after the other",
    "group_shapes": "print the problems of every grouped contraction launch issued outside a capture",
    "stage_profile": "print a cProfile of the staging thread (engine.StagedBatches) when a training loop's epoch ends",
}

# numeric / string knobs with a variable of their own (development A/B; defaults in the code that reads them)
KNOBS = {
    "EGK_LIB_PATH": "another build of libegopack_hip.so (A/B of two builds in one tree)",
    "EGK_WGRAD_COUNT": "bf16 weight-gradient problems per grouped launch (6)",
    "EGK_F32_WGRAD_COUNT": "exact-f32 weight-gradient problems per grouped launch (8)",
    "EGK_WGRAD_SCHED": "when parked weight gradients are issued: free | rows | inline",
    "EGK_WGRAD_KCHUNKS": "K pieces of a lone weight gradient",
    "EGK_LIVE_SHARE": "largest labelled share of a batch for which the heads are compacted (0.75)",
    "EGK_ROWS_V2": "the rows1024.h kernels (1)",
    "EGK_TRAIN_AFTER": "EgoPack step: the training pass starts behind this phase of the precise pass",
}

_seen_env = {}


def _names(var: str) -> dict:
    raw = os.environ.get(var, "")
    hit = _seen_env.get(var)
    if hit is not None and hit[0] == raw:
        return hit[1]
    out = {}
    for item in raw.split(","):
        item = item.strip()
        if item:
            k, _, v = item.partition("=")
            out[k] = v
    _seen_env[var] = (raw, out)
    unknown = [k for k in out if k not in (DEBUG if var == "EGK_DBG" else REGISTRY)]
    if unknown:
        warnings.warn(f"{var}: unknown switch name(s) {unknown} (see egopack_amd/switches.py)")
    return out


def enabled(name: str) -> bool:
    """The switch's value: its default unless EGK_DISABLE / EGK_ENABLE name it (EGK_DISABLE wins)."""
    default = REGISTRY[name][0]
    if name in _names("EGK_DISABLE"):
        return False
    if name in _names("EGK_ENABLE"):
        return True
    return default


def override(name: str):
    """True / False when the environment names the switch, None when it does not (per-object defaults: engine attributes)."""
    REGISTRY[name]
    if name in _names("EGK_DISABLE"):
        return False
    if name in _names("EGK_ENABLE"):
        return True
    return None


def value(name: str, default=None):
    """The ``=value`` an EGK_ENABLE entry carries (``name=4``), else ``default``."""
    REGISTRY[name]
    v = _names("EGK_ENABLE").get(name)
    return default if not v else v


def debug(name: str) -> bool:
    DEBUG[name]
    return name in _names("EGK_DBG")


def table() -> str:
    rows = [f"{'switch':26s} default  meaning"]
    for k, (d, m) in REGISTRY.items():
        rows.append(f"{k:26s} {'on ' if d else 'off'}      {m}")
    rows.append("")
    rows += [f"EGK_DBG={k:18s}          {m}" for k, m in DEBUG.items()]
    rows += [f"{k:26s}          {m}" for k, m in KNOBS.items()]
    return "\n".join(rows)


if __name__ == "__main__":
    print(table())
