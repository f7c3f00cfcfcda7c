"""ctypes binding of libegopack_hip.so -- the reference-side stub a maintainer adds.

The library is the product: if it is missing this module raises, there is no fallback of any
kind (the CPU oracle under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

# torch must be imported BEFORE the library is dlopen'ed: the PyTorch-ROCm wheel bundles its own HIP /
# HSA runtime; loading ours first would pull the system copies into the process and the second runtime
# then fails with "no ROCm-capable device is detected".  With torch first, the library's
# libamdhip64 dependency resolves to the runtime torch already loaded (one runtime, shared streams).
import torch  # noqa: F401

HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["EGK_LIB_PATH"]) if os.environ.get("EGK_LIB_PATH") else HERE / "libegopack_hip.so"  # (env: development A/B of two builds)
HEADER = HERE.parent / "include" / "egopack_hip.h"

vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float


class GemmDesc(C.Structure):
    """struct egk_gemm_desc (include/egopack_hip.h)."""
    _fields_ = [
        ("M", i32), ("N", i32), ("K1", i32), ("K2", i32),
        ("A1", vp), ("A2", vp), ("B1", vp), ("B2", vp),
        ("lda1", i64), ("lda2", i64), ("ldb1", i64), ("ldb2", i64),
        ("transA", i32), ("transB", i32), ("a_dtype", i32), ("b_dtype", i32),
        ("compute", i32),
        ("C", vp), ("ldc", i64), ("c_dtype", i32),
        ("accumulate", i32), ("act", i32), ("alpha", f32),
        ("bias", vp), ("residual", vp), ("ldr", i64), ("r_dtype", i32),
        ("splitk", i32), ("ws", vp), ("ws_bytes", i64), ("dbias", vp),
        ("st_mode", i32), ("st_nseg", i32), ("st_min_seg_rows", i32), ("st_seg_ptr", vp), ("st_ws", vp), ("st_x", vp),
        ("st_ldx", i64), ("st_stats", vp), ("st_w", vp), ("st_b", vp), ("st_slope", f32),
        ("n_extra", i32), ("xK", i32 * 4), ("xA", vp * 4), ("xB", vp * 4), ("xlda", i64 * 4), ("xldb", i64 * 4),
        ("op_f16", i32),
    ]


class HostDataset(C.Structure):
    """struct egk_host_dataset (include/egopack_hip.h): the per-sample tables of a resident dataset."""
    _fields_ = [("T", i32), ("S", i32), ("train", i32), ("y_heads", i32), ("L", i64), ("y_elems", i64), ("heavy_in_launch", i64),
                ("live_share", C.c_double),
                ("y", vp), ("pos", vp), ("tau", vp), ("first", vp), ("vlen", vp), ("starts", vp), ("ends", vp),
                ("n_tmpl", i64), ("e_max", i64), ("h_max", i64), ("th_max", i64),
                ("t_e", vp), ("t_ei", vp), ("t_col", vp), ("t_tcol", vp), ("t_tw", vp), ("t_rp", vp), ("t_trp", vp), ("t_band", vp),
                ("t_nh", vp), ("t_nth", vp), ("t_hv", vp), ("t_thv", vp), ("t_dmax", vp), ("t_tdmax", vp)]


class HostBatch(C.Structure):
    """struct egk_host_batch (include/egopack_hip.h): the caller-allocated arrays of one batch + what the call reports."""
    _fields_ = [("E", i64), ("heavy_cap", i64), ("t_heavy_cap", i64), ("live_cap", i64),
                ("y", vp), ("pos", vp), ("batch", vp), ("ptr", vp), ("ptr32", vp), ("x_idx", vp), ("edge_index", vp),
                ("rowptr", vp), ("col", vp), ("t_rowptr", vp), ("t_col", vp), ("t_wgt", vp), ("band", vp), ("heavy", vp), ("t_heavy", vp),
                ("live_idx", vp), ("live_inv", vp), ("live_y", vp),
                ("n_heavy", i64), ("n_t_heavy", i64), ("n_live", i64), ("pos_min", i64), ("pos_max", i64),
                ("heavy_mode", i32), ("t_heavy_mode", i32), ("edge_cap", i64), ("live_ap_first", i64), ("live_ap_step", i64)]


class HostPart(C.Structure):
    """struct egk_host_part (include/egopack_hip.h): one task batch as the merged batch's builder reads it."""
    _fields_ = [("n_nodes", i64), ("E", i64), ("n_heavy", i64), ("n_t_heavy", i64), ("edge_stride", i64), ("pos_min", i64), ("pos_max", i64),
                ("pos", vp), ("edge_index", vp), ("rowptr", vp), ("col", vp), ("t_rowptr", vp), ("t_col", vp), ("t_wgt", vp), ("band", vp),
                ("heavy", vp), ("t_heavy", vp), ("heavy_mode", i32), ("t_heavy_mode", i32)]


class HostMerged(C.Structure):
    """struct egk_host_merged (include/egopack_hip.h)."""
    _fields_ = [("edge_cap", i64), ("pos", vp), ("edge_index", vp), ("rowptr", vp), ("col", vp), ("t_rowptr", vp), ("t_col", vp),
                ("t_wgt", vp), ("band", vp), ("heavy", vp), ("t_heavy", vp), ("seg_ptr", vp),
                ("n_nodes", i64), ("E", i64), ("min_seg_rows", i64), ("pos_min", i64), ("pos_max", i64),
                ("heavy_mode", i32), ("t_heavy_mode", i32)]


class CETask(C.Structure):
    """struct egk_ce_task (include/egopack_hip.h)."""
    _fields_ = [("logits", vp * 4), ("ld", i64 * 4), ("C", i32 * 4), ("pad", i32 * 4), ("dcol", i64 * 4), ("n_heads", i32),
                ("y", vp), ("y_stride", i64), ("loss", vp), ("dlogits", vp), ("ldd", i64), ("rows", i32), ("gscale", f32)]


# name -> (restype, argtypes); mirrors include/egopack_hip.h one to one
SIGNATURES = {
    "egk_version": (C.c_int, []),
    "egk_last_error": (C.c_char_p, []),
    "egk_prof_enable": (C.c_int, [C.c_int]),
    "egk_prof_reset": (C.c_int, []),
    "egk_prof_count": (C.c_int, []),
    "egk_prof_get": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(i64), C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "egk_stamp": (C.c_int, [vp, vp, i32]),
    "egk_tee_split_next": (C.c_int, [vp, vp, i64]),
    "egk_gemm": (C.c_int, [vp, C.POINTER(GemmDesc)]),
    "egk_gemm_stats_blocks": (C.c_int, [C.POINTER(GemmDesc)]),
    "egk_gemm_grouped": (C.c_int, [vp, C.POINTER(GemmDesc), i32]),
    "egk_gemm_ws_bytes": (i64, [C.POINTER(GemmDesc)]),
    "egk_gemm_splitk": (C.c_int, [i32, i32, i32, i32]),
    "egk_gemm_set_pipeline": (C.c_int, [i32]),
    "egk_colsum_ws_len": (C.c_int, [i32, i32]),
    "egk_colsum": (C.c_int, [vp, vp, i64, i32, i32, vp, i32, vp, i32]),
    "egk_rowln_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, f32, u64, u64, vp, i32]),
    "egk_rowln_bwd_ws_rows": (C.c_int, [i32]),
    "egk_rowln_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32]),
    "egk_ln_bwd_reduce": (C.c_int, [vp, vp, vp, vp, i32, i32, i32]),
    "egk_ln_bwd_reduce_multi": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32]),
    "egk_rowln_group_fwd": (C.c_int, [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, f32, i32, i32]),
    "egk_rowln_group_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32]),
    "egk_graphln_ws_bytes": (i64, [i32, i32, i32]),
    "egk_graphln_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32]),
    "egk_graphln_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32]),
    "egk_rowdot_ws_rows": (i32, [i32]),
    "egk_rowdot_bce": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32]),
    "egk_rowdot_reduce": (C.c_int, [vp, vp, vp, vp, i32, i32]),
    "egk_rowdot_ce2_max_rows": (i32, []),
    "egk_rowdot_ce2": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32]),
    "egk_rowdot_ce2_multi": (C.c_int, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, i32]),
    "egk_graphln_stats_blocks": (i32, [i32]),
    "egk_graphln_stats": (C.c_int, [vp, vp, vp, i32, i32, i32, vp, i32]),
    "egk_graphln_bwd_stats": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, i32]),
    "egk_graphln_bwd_finish": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, vp, i32]),
    "egk_graphln_fwd_apply": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32]),
    "egk_graphln_bwd_apply": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, vp, i32]),
    "egk_pe_table": (C.c_int, [vp, vp, i64, i32, i32, vp]),
    "egk_pe_add_table": (C.c_int, [vp, vp, vp, vp, vp, i64, i32, vp, i32, i32, i32]),
    "egk_pe_add": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32]),
    "egk_csr_gather": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, i32]),
    "egk_csr_gather_banded": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp, i32]),
    "egk_csr_heavy_ws_bytes": (i64, [i32, i32]),
    "egk_csr_heavy_threshold": (i32, []),
    "egk_gather_max_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "egk_gather_max_group_fwd": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32]),
    "egk_gather_max_tune": (C.c_int, [i32]),
    "egk_gather_max_bwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "egk_segment_max_fwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32]),
    "egk_segment_max_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "egk_segment_max_multi_fwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "egk_segment_max_multi_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "egk_row_inv_norm": (C.c_int, [vp, vp, vp, i32, i32, i32]),
    "egk_cos_dist": (C.c_int, [vp, vp, i64, vp, vp, vp, i32, i32]),
    "egk_topk_smallest": (C.c_int, [vp, vp, i64, vp, vp, vp, i32, i32, i32]),
    "egk_row_sq_norm": (C.c_int, [vp, vp, vp, i32, i32, i32]),
    "egk_topk_smallest_l2": (C.c_int, [vp, vp, i64, vp, vp, vp, i32, i32, i32]),
    "egk_topk_window": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "egk_gemm_defer_reduce_next": (C.c_int, [i32]),
    "egk_slab_input_next": (C.c_int, [vp, vp, vp]),
    "egk_gemm_reduce_slabs": (C.c_int, [vp, vp, i32, i32, i32, vp, vp, i64]),
    "egk_topk_window_group": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32]),
    "egk_topk_window_group16": (C.c_int, [vp, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32]),
    "egk_residual_ratio16": (C.c_int, [vp, vp, i64, vp, vp, i32, i32, i32]),
    "egk_cast_f16": (C.c_int, [vp, vp, vp, i64]),
    "egk_row_inv_norm_cast": (C.c_int, [vp, vp, vp, vp, vp, i32, i32]),
    "egk_bf16_residual_ratio": (C.c_int, [vp, vp, i64, vp, vp, i32, i32]),
    "egk_gather_max_bank_grad": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32]),
    "egk_segment_sum_rows_f64": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i32]),
    "egk_onehot_sigmoid_loss_fwd": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, f32, f32]),
    "egk_onehot_sigmoid_loss_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, i32]),
    "egk_ce_fwd": (C.c_int, [vp, vp, i64, vp, i64, vp, vp, i32, i32, f32, i32]),
    "egk_ce_bwd": (C.c_int, [vp, vp, i64, vp, i64, vp, vp, vp, i64, i32, i32, f32, i32]),
    "egk_ce_fused_multi": (C.c_int, [vp, vp, i32, f32, i32]),
    "egk_ce_fused": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, i64, vp, vp, i64, i32, f32, f32, i32]),
    "egk_bce_fwd": (C.c_int, [vp, vp, vp, vp, i32]),
    "egk_bce_bwd": (C.c_int, [vp, vp, vp, vp, vp, i32, i32]),
    "egk_dropout_fwd": (C.c_int, [vp, vp, vp, vp, i64, f32, u64, u64, vp, i32]),
    "egk_dropout_bwd": (C.c_int, [vp, vp, vp, vp, i64, f32, i32]),
    "egk_relu_gate": (C.c_int, [vp, vp, vp, vp, i64, i32]),
    "egk_cast": (C.c_int, [vp, vp, i32, vp, i32, i64]),
    "egk_split_bf16": (C.c_int, [vp, vp, i64, vp, vp, i64, i64, i64]),
    "egk_host_bounded_draws": (i64, [vp, vp, vp, i64, i32, vp]),
    "egk_host_window_rows": (i64, [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "egk_host_build_batch": (i64, [C.POINTER(HostDataset), vp, vp, vp, i64, C.POINTER(HostBatch)]),
    "egk_host_batch_sizes": (i64, [C.POINTER(HostDataset), vp, i64, vp]),
    "egk_host_merge_batches": (i64, [C.POINTER(HostPart), i32, C.POINTER(HostMerged)]),
    "egk_tune": (C.c_int, [i32, i32]),
    "egk_weighted_sums": (C.c_int, [vp, vp, vp, vp, i32, vp]),
    "egk_weighted_sums_acc": (C.c_int, [vp, vp, vp, vp, i32, vp, vp]),
    "egk_fill_scaled_multi": (C.c_int, [vp, vp, vp, vp, vp, i32]),
    "egk_copy_blocks": (C.c_int, [vp, vp, vp, vp, i32]),
    "egk_gather_rows": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, i32, i64, i32]),
    "egk_gather_lerp_rows": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, vp, vp, i32, i64, i32]),
    "egk_label_rank": (C.c_int, [vp, vp, i64, vp, i64, vp, i32, i32]),
    "egk_edit_distance": (C.c_int, [vp, vp, i64, i64, i64, vp, i64, i64, vp, i32, i32, i32]),
    "egk_cast_rows": (C.c_int, [vp, vp, i32, i64, vp, i32, i64, i32, i32, i32]),
    "egk_axpby": (C.c_int, [vp, vp, vp, vp, i64, f32, f32]),
    "egk_fill_scaled": (C.c_int, [vp, vp, f32, vp, i64]),
    "egk_sum_scale": (C.c_int, [vp, vp, vp, i64, f32, i32]),
    "egk_adam_step": (C.c_int, [vp, vp, vp, i32, vp, vp, i64, vp, f32, f32, f32, f32, vp]),
    "egk_zero_fill": (C.c_int, [vp, vp, i64]),
    "egk_zero_fill_ranges": (C.c_int, [vp, vp, vp, vp, i32]),
    "egk_adam_step_bump": (C.c_int, [vp, vp, vp, i32, vp, vp, i64, vp, f32, f32, f32, f32, vp, vp, vp, i64]),
    "egk_adam_hyper": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, vp]),
}


def header_symbols() -> list:
    """Every function name declared in include/egopack_hip.h."""
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egk_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load() -> C.CDLL:
    """Load the library (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m egopack_amd.build` "
            "(or __graft_entry__.build()).  egopack_amd has no CPU / eager fallback.")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().egk_last_error().decode(errors="replace")
