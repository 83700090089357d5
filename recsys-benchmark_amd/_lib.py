"""ctypes binding of libmi355x_recsys.so (the C-ABI in include/mi355x_recsys.h).

The library is the product: there is NO CPU or eager-PyTorch fallback behind
these calls.  If the shared object has not been built, or a tensor is not on a
ROCm device, the caller gets a loud error (never a silent slow path).
"""
import ctypes
import os
import threading
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355x_recsys.so")

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32

# name -> argtypes; every entry point returns int unless listed in _RESTYPES.
# tests/test_abi.py cross-checks this table against include/mi355x_recsys.h.
ABI_VERSION = 3          # MI_ABI_VERSION of include/mi355x_recsys.h these bindings were written against

SIGNATURES = {
    "mi_abi_version": [],
    "mi_strerror": [ctypes.c_int],
    "mi_gather_fm_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _p],
    "mi_gather_fm_fwd_ld": [_p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _p],
    "mi_gather_fm_fwd_sum": [_p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _p],
    "mi_gather_fm_fwd_ride": [_p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _p, _p],
    "mi_prefetch_rows": [_p, _p, _p, _i64, _p, _i64, _i64, _i32, _i64, _p],
    "mi_gather_fm_bwd_rows": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _p],
    "mi_gather_fm_bwd_dense": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p],
    "mi_gather_rows_fwd": [_p, _p, _p, _i64, _i32, _i64, _p, _p],
    "mi_scatter_add_rows": [_p, _p, _p, _i64, _i32, _i64, _p],
    "mi_fm_fwd": [_p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _p, _p],
    "mi_dual_gather_fwd": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _i64, _i64, _i64,
                           _i32, _i32, _p, _p],
    "mi_dual_gather_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _i64,
                           _i64, _i64, _i32, _i32, _p],
    "mi_dual_gather_bwd_rows_workspace_elems": [_i32, _i64],
    "mi_dual_gather_bwd_rows_overwrites": [_i32, _i64],
    "mi_dual_gather_fwd_off": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _i32, _p, _p],
    "mi_dual_gather_bwd_rows": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i32, _p, _p],
    "mi_dual_gather_bwd_fields": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i64, _i64,
                                  _i64, _i64, _i32, _i32, _p, _i32, _p, _p, _p],
    "mi_xform_gather_fwd": [_p, _p, _p, _p, _i64, _i64, _p, _i64, _i32, _i64, _i32, _p, _p],
    "mi_xform_gather_bwd": [_p, _p, _p, _p, _p, _i64, _i64, _p, _p, _i64, _i64, _i32, _i64, _i32, _p],
    "mi_gather_rows_quant": [_p, _p, _i32, _p, _p, _p, _i64, _i32, _i64, _p, _p],
    "mi_csr_rows_fwd": [_p, _p, _p, _p, _p, _i64, _i32, _i64, _p, _p],
    "mi_dhe_hash": [_p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _p],
    "mi_spmm_csr": [_p, _p, _p, _p, _p, _i32, _p, _p, _p, _i32, _p, ctypes.c_float, _i32, _i32, _p, _i32, _p, _i32, _p],
    "mi_spmm_csr_masked": [_p, _p, _p, _p, _p, _i32, _p, _p, _p, _i32, _p, ctypes.c_float, _i32, _i32, _p, _i32, _p, _i32, _p,
                           _p],
    "mi_spmm_csr_sel": [_p, _p, _p, _p, _p, _i32, _p, _p, _p, _i32, _p, ctypes.c_float, _i32, _i32, _p, _i32, _p, _i32, _p,
                        _p, _p],
    "mi_batch_row_list": [_p, _p, _p, _i64, _i64, _i64, _p, _i32, _p, _p, _p],
    "mi_row_mask": [_p, _p, _i32, _i32, _i32, _p, _p],
    "mi_spmm_sliced": [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _p, _p, _p, _i32, _p, ctypes.c_float, _i32, _p, _i32,
                       _p, _p],
    "mi_spmm_tiled": [_p, _p, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p, _p, _i32, _p, ctypes.c_float, _i32, _p],
    "mi_gemm_f32": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _i64,
                    _i64, _i32, _p, _p, _i32, _i64, _p, _i32, _i64, _p, _i32, _p, _i32, _i64, _i32, _p],
    "mi_cross_bwd_pre": [_p, _p, _p, _p, _p, _i64, _i32, _p],
    "mi_colsum": [_p, _i32, _p, _i32, _p, _p, _i32, _i32, _p],
    "mi_rowdot": [_p, _i32, _p, _p, _p, _p, _i32, _i32, _p],
    "mi_bce_logits_fwd": [_p, _p, _p, _p, _i64, _p],
    "mi_bce_logits_bwd": [_p, _p, _p, _p, _i64, _p],
    "mi_outer": [_p, _p, _p, _i32, _i32, _p],
    "mi_mix_gate_bwd": [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p],
    "mi_tt_fwd": [_p, _p, _i32, _p, _p, _p, _p, _i64, _i32, _i64, _p, _p],
    "mi_tt_bwd": [_p, _p, _p, _p, _i32, _p, _p, _p, _i64, _i32, _i64, _p],
    "mi_bn_relu_dropout_fwd": [_p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, ctypes.c_float, ctypes.c_float,
                               ctypes.c_float, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p, _p],
    "mi_bn_relu_dropout_bwd": [_p, _p, _i32, _i32, _i32, _i32, _i32, _p, ctypes.c_float, _p, _p, _p, _p, _p, _p, _p, _p,
                               _p],
    "mi_sparse_adam_sorted": [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i64, ctypes.c_float, _p, ctypes.c_double,
                              ctypes.c_double, ctypes.c_float, _p],
    "mi_sparse_adam_sorted_ld": [_p, _p, _p, _p, _i64, _p, _p, _p, _i64, _i32, _i64, ctypes.c_float, _p, ctypes.c_double,
                                 ctypes.c_double, ctypes.c_float, _p],
    "mi_coalesce_rows_sorted": [_p, _p, _p, _p, _p, _i64, _i32, _i64, _p],
    "mi_sort_field_rows_workspace_bytes": [_i64, _i32],
    "mi_sort_field_rows": [_p, _p, _i64, _i64, _i32, _p, _p, _p, _p, _p],
    "mi_adam_dense_multi": [_p, _p, _p, _p, _p, _p, _i32, ctypes.c_float, ctypes.c_double, ctypes.c_double, ctypes.c_float,
                            ctypes.c_float, _p, _p],
    "mi_adam_tick": [_p, _p, ctypes.c_double, ctypes.c_double, ctypes.c_double, _p],
    "mi_adam_tick_multi": [_p, _p, _i32, ctypes.c_double, ctypes.c_double, ctypes.c_double, _p],
    "mi_scatter_axpy_rows": [_p, _p, ctypes.c_float, _p, _i64, _i32, _i64, _p],
    "mi_tt_digits": [_p, _i64, _i64, _p, _i32, _p, _p, _p, _p],
    "mi_move_chunks": [_p, _p, _i64, _p, _p, _i64, _i32, _i64, _p, _i32, _p],
    "mi_gemm_f32_row_groups": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _p, _p],
    "mi_gemm_f32_k_groups": [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i64, _p, _i32, _p],
    "mi_tt_last_fwd": [_p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _p, _i64, _p],
    "mi_tt_last_bwd": [_p, _p, _i64, _p, _p, _p, _i32, _i32, _i32, _p, _p, _p, _i64, _p, _p, _i32, _p, _i64, _p],
    "mi_tt_plan_level": [_p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _i32, _p],
    "mi_segment_sum": [_p, _i64, _i32, _p, _i32, _p, _i64, _p],
    "mi_rowsq_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p],
    "mi_rowsq_fwd_armed": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p],
    "mi_rowsq_bwd": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p, _p],
    "mi_bpr_workspace_elems": [_i64],
    "mi_bpr_fwd": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p, _p],
    "mi_bpr_fwd_armed": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p, _p],
    "mi_bpr_fwd_plus": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _i32, _p, ctypes.c_float, _p, _p],
    "mi_bpr_bwd": [_p, _p, _p, _p, _p, _p, _i64, _i32, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p],
    "mi_mask_topk_rows": [_p, _i64, _i64, _i64, _p, _p, _p, _i32, _p, _p, _p],
    "mi_rownorm_fwd": [_p, _i64, _i32, ctypes.c_float, _p, _p, _p],
    "mi_rownorm_bwd": [_p, _p, _p, _i64, _i32, ctypes.c_float, _p, _p],
    "mi_lse_diag_workspace_elems": [_i32],
    "mi_lse_diag_fwd": [_p, _i64, _i32, ctypes.c_float, _p, _p, _p, _p, _p, _p],
    "mi_lse_diag_bwd": [_p, _i64, _i32, ctypes.c_float, _p, _p, _p, _p, _i32, _p],
    "mi_qat_gather_fwd": [_p, _p, _p, _i32, _p, _p, _i64, _p, _i64, _i32, _i64, _p, _p],
    "mi_qat_gather_bwd": [_p, _p, _p, _i32, _p, _p, _i64, _p, _p, _p, _i64, _i32, _i64, _p],
    "mi_optembed_fwd": [_p, _p, _p, _p, _i32, _p, _i32, _p, _i64, _i32, _i64, _p, _p],
    "mi_optembed_bwd": [_p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _i64, _p],
    "mi_route_workspace_elems": [_i64, _i32],
    "mi_route_buckets": [_p, _p, _i64, _i32, _i32, _i64, _i64, _p, _p, _p, _p, _p, _p],
    "mi_unpack_rows": [_p, _p, _p, _i64, _i32, _p],
    "mi_gather_pack_rows": [_p, _p, _p, _p, _i64, _i32, _i64, _p, _p],
    "mi_slot_fm_fwd": [_p, _p, _i64, _p, _p, _p, _i64, _i32, _i32, _p, _p],
    "mi_slot_fm_bwd": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i32, _i32, _p],
    "mi_prof_enable": [_i32],
    "mi_col_act_fwd": [_p, _i32, _p, _p, _p, _i32, _p, _p, _i32, _i32, _p],
    "mi_col_act_bwd": [_p, _p, _i32, _p, _p, _p, _i32, _p, _p, _i32, _i32, _p],
    "mi_bn_dz": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _i32, _p],
    "mi_bn_mish_bwd": [_p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _p],
    "mi_tail_dropout_masks": [_p, _i32, _p, _p, _p, _p, _i32, _p],
    "mi_tail_dropout_masks_z": [_p, _i32, _p, _p, _p, _p, _i32, _p, _i64, _p],
    "mi_tail_fwd_gemm": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _i32, _p, _i32, _p, _p, _i32, _i32, _i32, _p],
    "mi_tail_fwd_gemm_m": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _i32, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p],
    "mi_tail_fwd_gemm_head": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _i32, _i32, _i32, _i32, _p, _p, _p],
    "mi_tail_fwd_gemm_s": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _i32, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _i32, _p, _p, _p, _p],
    "mi_tail_dgrad_gemm_s": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, ctypes.c_float, _p, _p,
                             _i32, _p, _i32, _p, _i32, _i32, _i32, _p, _p],
    "mi_tail_head_fwd_m": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p, _i32, _i32, _p, _p],
    "mi_tail_dgrad_gemm_m": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, ctypes.c_float, _p, _p,
                             _i32, _p, _p, _i32, _i32, _i32, _p, _p],
    "mi_tail_dgrad_gemm_fm": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p],
    "mi_tail_dgrad_gemm_fm_slot": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i32, _p],
    "mi_tail_part_elems": [_i32, _i32],
    "mi_tail_bn_finalize_fwd": [_p, _i32, _i32, _p, _p, _p, _p, _p, ctypes.c_float, ctypes.c_float, _p, _p, _p, _p, _p,
                                _p, _p],
    "mi_tail_bn_finalize_fwd_r": [_p, _i32, _i32, _p, _p, _p, _p, _p, ctypes.c_float, ctypes.c_float, _p, _p, _p, _p, _p,
                                  _p, _p, _p],
    "mi_tail_head_fwd": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p, _i32, _i32, _p],
    "mi_tail_head_blocks": [_i32],
    "mi_tail_head_bwd": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p, _p, _i32, _i32, _p],
    "mi_tail_head_bwd_s": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _p],
    "mi_tail_head_bce_ws_elems": [_i32],
    "mi_tail_head_bce": [_p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _i32, _i32, _p, _p, _p],
    "mi_tail_bn_finalize_bwd": [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p],
    "mi_tail_affine_consts": [_i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "mi_tail_bn_finalize_bwd_a": [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _p, _p],
    "mi_tail_bn_finalize_bwd_b": [_p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _i32, _p, _p, _p],
    "mi_tail_dgrad_gemm": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _i32, _p, _p, _p, ctypes.c_float, _p, _p,
                           _i32, _p, _p, _i32, _i32, _i32, _p],
    "mi_tail_wgrad_splits": [_i32, _i32, _i32],
    "mi_tail_wgrad_gemm": [_p, _p, _i32, _p, _p, _p, _p, _p, _i32, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i32,
                           _i32, _i32, _p],
    "mi_gemm_f32_multi": [_p, _i32, _i32, _i32, _p],
    "mi_gemm_f32_multi_ride": [_p, _i32, _i32, _i32, _p, _p],
    "mi_gemm_f32_multi_plan": [_p, _i32, _i32, _i32, _p, _p, _p],
    "mi_gemm_f32_panel": [_p, _i32, _p, _i32, _i32, _i32, _i64, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i32, _p, _p],
    "mi_mix_expert_fwd": [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "mi_mix_expert_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _p],
    "mi_rowdot_multi": [_p, _i32, _p, _p, _i32, _i32, _i32, _p],
    "mi_cross_bwd_head": [_p, _p, _p, _p, _i32, _p, _p, _p, _i32, _p, _p, _i32, _i32, _p],
    "mi_comm_available": [],
    "mi_comm_unique_id": [ctypes.c_char_p],
    "mi_comm_init": [ctypes.c_char_p, _i32, _i32, ctypes.POINTER(ctypes.c_void_p)],
    "mi_comm_destroy": [_p],
    "mi_comm_abort": [_p],
    "mi_comm_all_to_all": [_p, _p, _p, _i64, _p],
    "mi_comm_all_reduce_sum_f32": [_p, _p, _i64, _p],
    "mi_prof_count": [],
    "mi_prof_empty_launch": [_i32, _i32, _p],
    "mi_prof_read": [_i32, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)],
}
_RESTYPES = {"mi_strerror": ctypes.c_char_p, "mi_route_workspace_elems": ctypes.c_int64,
             "mi_bpr_workspace_elems": ctypes.c_int64, "mi_lse_diag_workspace_elems": ctypes.c_int64,
             "mi_sort_field_rows_workspace_bytes": ctypes.c_int64, "mi_tail_part_elems": ctypes.c_int64,
             "mi_dual_gather_bwd_rows_workspace_elems": ctypes.c_int64}

_lib: Optional[ctypes.CDLL] = None
_lock = threading.Lock()


class MI355XLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise MI355XLibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C recsys-benchmark_amd/csrc`). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        if lib.mi_abi_version() != ABI_VERSION:
            raise MI355XLibraryError("libmi355x_recsys.so ABI version mismatch; rebuild it")
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mi_strerror(rc).decode()
        raise MI355XLibraryError(f"{what or 'mi355x_recsys call'} failed: {msg} ({rc})")


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(*tensors: torch.Tensor) -> torch.device:
    """All tensors must live on one ROCm device; the kernels have no CPU path."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise MI355XLibraryError(
                "recsys_benchmark_amd runs its hot path as HIP kernels on an MI355X; got a tensor on "
                f"'{t.device}'. Move the module and its inputs to 'cuda' (there is no CPU fallback)."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise MI355XLibraryError(f"tensors on different devices: {dev} vs {t.device}")
    assert dev is not None
    return dev


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


# ---- sticky out-of-range flag (one int32 word per device) ---------------------
_err_words: Dict[int, torch.Tensor] = {}


def err_word(device: torch.device) -> torch.Tensor:
    i = device.index if device.index is not None else torch.cuda.current_device()
    w = _err_words.get(i)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", i))
        _err_words[i] = w
    return w


def check_index_errors(device: Optional[torch.device] = None) -> None:
    """Synchronise and raise IndexError if any lookup since the last check was out of range.

    Mirrors what nn.Embedding does eagerly on CPU in the reference
    (src/models/embeddings/base.py:74-75); here it is deferred so the hot path never syncs.
    """
    words = _err_words.values() if device is None else [err_word(device)]
    for w in words:
        bits = int(w.item())
        if bits != 0:
            w.zero_()
            if bits & 1:
                raise IndexError("index out of range in embedding lookup (mi355x_recsys)")
            raise IndexError("an id lies outside its own field's range (inside the concatenated table): the lookup read "
                             "another field's row like the reference does, but the field-sorted sparse optimizer dropped "
                             "its gradient (mi355x_recsys)")
