"""ctypes binding of libgte_hip.so (the C ABI declared in include/gte.h).

The product path has NO fallback: if the shared library is missing, or a compute
entry point is called without a HIP device, this raises -- loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTE_LIB_PATH: profiling builds of the same ABI (e.g. ablation variants); default = the in-tree library
LIB_PATH = os.environ.get("GTE_LIB_PATH") or os.path.join(_HERE, "libgte_hip.so")

# name -> (restype, argtypes); mirrors include/gte.h declaration by declaration
SIGNATURES = {
    "gte_version": (c_int, []),
    "gte_last_error": (c_char_p, []),
    "gte_device_info": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "gte_spmm_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                             c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_spmm_csr_accumulate": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                        c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_spmm_csr_edge_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_spmm_csr_edge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int,
                                  c_void_p, c_int64, c_void_p]),
    "gte_spmm_tile_rows": (c_int, []),
    "gte_spmm_csr_tiled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_coo_to_csr_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_coo_to_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_batch_csr": (c_int, [c_void_p, c_int64] + [c_void_p] * 10 + [c_int64, c_int64, c_void_p]),
    "gte_cast_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "gte_gemm_bf16_nt": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p]),
    "gte_knn_max_k": (c_int, []),
    "gte_knn_max_page_nodes": (c_int, []),
    "gte_knn_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "gte_visibility_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "gte_knn_csr_workspace_bytes": (c_int64, [c_int64]),
    "gte_knn_csr": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_int64, c_void_p]),
    "gte_island_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_batch_assemble": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_batch_assemble_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                        c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "gte_batch_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                               c_int64, c_void_p]),
    "gte_edge_weights_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_edge_weights_bbox": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                      c_int64, c_void_p]),
    "gte_bbox_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "gte_inv_degree": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_sage_linear_fwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "gte_sage_linear_fwd_fuses_ln": (c_int, [c_int64, c_int64]),
    "gte_sage_linear_fwd_p3": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                       c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                       c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "gte_sage_linear_dw_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int64]),
    "gte_sage_linear_dw": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                   c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_spmm_csr_accumulate_ln_supported": (c_int, [c_int64]),
    "gte_spmm_csr_accumulate_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                           c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "gte_gemm_tail_workspace_bytes": (c_int64, []),
    "gte_gemm_set_tail_workspace": (c_int, [c_void_p, c_int64]),
    "gte_gemm_set_mode": (c_int, [c_int]),
    "gte_gemm_get_mode": (c_int, []),
    "gte_gemm_set_thread_mode": (c_int, [c_int]),
    "gte_p3_row_bytes": (c_int64, [c_int64]),
    "gte_p3_from_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "gte_p3_to_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_p3_from_f32_batch": (c_int, [c_void_p, c_int, c_void_p]),
    "gte_spmm_csr_p3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "gte_spmm_csr_accumulate_ln_p3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64,
                                              c_int, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p, c_int64,
                                              c_void_p, c_void_p]),
    "gte_ln_relu_bwd_p3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_void_p, c_int64, c_void_p]),
    "gte_sage_smallk_bwd_supported": (c_int, [c_int64, c_int64]),
    "gte_sage_smallk_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "gte_sage_smallk_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                    c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_nt": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                               c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_sage_narrow_bwd_ln_p3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                                          c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                          c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_sage_narrow_pad_supported": (c_int, [c_int64, c_int64, c_int64]),
    "gte_sage_narrow_fwd_pad": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                        c_void_p, c_int64, c_int64, c_void_p]),
    "gte_sage_narrow_bwd_ln_p3_pad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                              c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                              c_int64, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_nt_ln_bwd_supported": (c_int, [c_int64]),
    "gte_gemm_p3_nt_ln_bwd_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_gemm_p3_nt_ln_bwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_nt_smallk_bwd_supported": (c_int, [c_int64, c_int64]),
    "gte_gemm_p3_nt_smallk_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "gte_gemm_p3_nt_smallk_bwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                          c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                          c_void_p]),
    "gte_gemm_p3_nt_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_gemm_p3_tn_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                    c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_nt_rows2": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                     c_int64, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "gte_gemm_p3_tn_rows2": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                     c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_nt_ln_fwd_supported": (c_int, [c_int64]),
    "gte_gemm_p3_nt_ln_fwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                      c_int64, c_int64, c_void_p]),
    "gte_gemm_p3_nt_rows2_ln_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                            c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                            c_void_p, c_int64, c_int64, c_void_p]),
    "gte_gemm_p3_set_rows64": (c_int, [c_int]),
    "gte_gemm_p3_nt_plan": (c_int, [c_int64, c_int64, c_int64, c_int64, c_int, c_int, c_int, ctypes.POINTER(c_int)]),
    "gte_gemm_p3_set_nt_cfg": (c_int, [c_int]),
    "gte_gemm_p3_set_ln_rows": (c_int, [c_int]),
    "gte_batch_assemble_defer": (c_int, [c_int]),
    "gte_head_dlq_finish_workspace_bytes": (c_int64, [c_int64]),
    "gte_head_dlq_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_float, c_void_p,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_gemm_p3_tn_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64, c_int64]),
    "gte_gemm_p3_tn": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                               c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_adam_ticket_bytes": (c_int64, []),
    "gte_gcnsage_step": (c_int, [c_void_p, c_int, POINTER(c_int), c_void_p]),
    "gte_gcnsage_forward": (c_int, [c_void_p, c_void_p]),
    "gte_fold_defer_begin": (c_int, [c_void_p]),
    "gte_fold_defer_flush": (c_int, []),
    "gte_fold_defer_flush_adam": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "gte_fold_defer_flush_adam_images": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_int, POINTER(c_int)]),
    "gte_sage_transform_fwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                       c_int64, c_void_p]),
    "gte_sage_qform_dw_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "gte_sage_qform_dw": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                  c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_sage_qform_dx": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                  c_int64, c_int64, c_void_p]),
    "gte_sage_narrow_supported": (c_int, [c_int64, c_int64]),
    "gte_sage_narrow_fwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p, c_int64, c_int64, c_void_p]),
    "gte_sage_narrow_fwd_ln_supported": (c_int, [c_int64, c_int64]),
    "gte_sage_narrow_fwd_ln": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                                       c_void_p]),
    "gte_sage_narrow_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "gte_sage_narrow_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                    c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                    c_void_p]),
    "gte_head_supported": (c_int, [c_int64, c_int64]),
    "gte_head_agg_ce_workspace_bytes": (c_int64, [c_int64]),
    "gte_head_agg_ce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "gte_sage_narrow_bwd_ce": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                       c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                       c_void_p, c_float, c_void_p, c_void_p]),
    "gte_sage_narrow_bwd_ln_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_ln_relu_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p,
                                c_int64, c_int64, c_void_p]),
    "gte_ln_relu_fwd_p3": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_int, c_void_p, c_int64, c_void_p, c_int64,
                                   c_void_p, c_int64, c_int64, c_void_p]),
    "gte_colsum_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_colsum": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_ln_relu_bwd_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "gte_ln_relu_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                c_void_p, c_int64, c_void_p]),
    "gte_gemm_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "gte_gemm_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                             c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "gte_gat_scores": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                               c_int, c_int, c_void_p]),
    "gte_gat_aggregate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "gte_gat_aggregate_fwd_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_int64,
                                         c_void_p, c_int64, c_void_p, c_void_p]),
    "gte_gat_dout_prepare": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                     c_void_p]),
    "gte_gat_bwd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "gte_gat_aggregate_bwd": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_void_p, c_int64] + [c_void_p] * 7 +
                              [c_int64] + [c_void_p] * 4 + [c_int64] + [c_void_p] * 3 + [c_int64, c_int, c_int,
                                                                                        c_void_p, c_int64, c_void_p]),
    "gte_gat_aggregate_bwd_ex": (c_int, [c_void_p] * 6 + [c_int64, c_int, c_void_p, c_int64] + [c_void_p] * 7 +
                                 [c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64] + [c_void_p] * 4 + [c_int64] +
                                 [c_void_p] * 3 + [c_int64, c_int, c_int, c_void_p, c_int64, c_void_p]),
    "gte_weighted_ce_workspace_bytes": (c_int64, [c_int64]),
    "gte_weighted_ce": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int, c_float,
                                c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "gte_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float,
                              c_float, c_float, c_int64, c_float, c_void_p]),
    "gte_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gte_adam_step_dev_images": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p]),
}

class P3Desc(ctypes.Structure):
    """gte_p3_desc of include/gte.h"""
    _fields_ = [("src", c_void_p), ("ld", c_int64), ("rows", c_int64), ("cols", c_int64), ("transpose", c_int),
                ("dst", c_void_p), ("ldp", c_int64)]


class StepLayer(ctypes.Structure):
    """gte_step_layer of include/gte.h"""
    _fields_ = [("kind", c_int), ("fin", c_int64), ("fout", c_int64),
                ("W", c_void_p), ("bias", c_void_p), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("relu", c_int),
                ("gW", c_void_p), ("gbias", c_void_p), ("ggamma", c_void_p), ("gbeta", c_void_p),
                ("wimg_fwd", c_void_p), ("ldp_wfwd", c_int64), ("wimg_bwd", c_void_p), ("ldp_wbwd", c_int64),
                ("x", c_void_p), ("ldx", c_int64), ("hp", c_void_p), ("ldp_h", c_int64), ("h_rows", c_void_p), ("n_res_rows", c_int64),
                ("make_hp", c_int),
                ("ahn", c_void_p), ("t", c_void_p), ("stats", c_void_p), ("y", c_void_p), ("yp", c_void_p), ("ldp_y", c_int64),
                ("dy", c_void_p), ("dzp", c_void_p), ("qp", c_void_p), ("ldp_o", c_int64),
                ("ws_ln", c_void_p), ("ws_ln_bytes", c_int64), ("ws_dw", c_void_p), ("ws_dw_bytes", c_int64),
                ("ldf", c_int64), ("ahnp", c_void_p), ("ldp_ahn", c_int64)]


class StepPlan(ctypes.Structure):
    """gte_step_plan of include/gte.h"""
    _fields_ = [("n_hidden", c_int), ("layer", StepLayer * 7),
                ("out_fin", c_int64), ("n_classes", c_int64),
                ("W_out", c_void_p), ("b_out", c_void_p), ("gW_out", c_void_p), ("gb_out", c_void_p),
                ("h_out", c_void_p), ("ld_h_out", c_int64),
                ("logits", c_void_p), ("tn", c_void_p), ("q_out", c_void_p), ("dl", c_void_p), ("dh_out", c_void_p),
                ("ce_part", c_void_p), ("ce_part_bytes", c_int64), ("ws_nar", c_void_p), ("ws_nar_bytes", c_int64),
                ("indptr", c_void_p), ("indices", c_void_p), ("w_in", c_void_p),
                ("rindptr", c_void_p), ("rindices", c_void_p), ("w_out", c_void_p),
                ("n_nodes", c_int64),
                ("labels", c_void_p), ("labels_f32", c_int), ("class_weights", c_void_p), ("grad_scale", c_float), ("out3", c_void_p),
                ("wimg_descs", c_void_p), ("n_wimg_descs", c_int),
                ("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n_param", c_int64),
                ("hyper", c_void_p), ("step_counter", c_void_p), ("ticket", c_void_p),
                ("tail_ws", c_void_p), ("tail_ws_bytes", c_int64), ("fuse_ln_dx", c_int),
                ("wimg_fresh", c_int), ("wimg_in_fold", c_int),
                ("out_gemm", c_int), ("ld_lg", c_int64), ("hp_out", c_void_p), ("ldp_hout", c_int64),
                ("wimg_out_fwd", c_void_p), ("ldp_wout_fwd", c_int64), ("wimg_out_bwd", c_void_p), ("ldp_wout_bwd", c_int64),
                ("dlqp", c_void_p), ("ldp_dlq", c_int64), ("ws_out", c_void_p), ("ws_out_bytes", c_int64),
                ("ws_ce", c_void_p), ("ws_ce_bytes", c_int64), ("ws_cs", c_void_p), ("ws_cs_bytes", c_int64),
                ("fwd_events", c_void_p)]


class BatchArrays(ctypes.Structure):
    """gte_batch_arrays of include/gte.h (one CSR direction of gte_batch_assemble)"""
    _fields_ = [(n, c_void_p) for n in ("edge_off", "indptr_loc", "indices_loc", "weight", "b_edge_off", "indptr_out",
                                        "indices_out", "weight_out")]


GTE_F32, GTE_BF16 = 0, 1
REDUCE_SUM, REDUCE_MEAN = 0, 1

_lib = None


class GteError(RuntimeError):
    pass


def load():
    """Load libgte_hip.so once; raise if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GteError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C gnn-tableextraction_amd/csrc`. There is no CPU fallback for the HIP path.")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gte_last_error()
        raise GteError(f"{what or 'gte call'} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_device(t, what: str) -> None:
    if not t.is_cuda:
        raise GteError(f"{what}: tensor is on {t.device}; the HIP path needs device tensors "
                       f"(there is no CPU fallback -- the CPU oracle lives under oracle/ for tests only)")
