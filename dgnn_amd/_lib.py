"""ctypes binding of libdgnn_hip.so (the C ABI declared in include/dgnn_hip.h).

There is NO fallback: if the library has not been built (`python -c "import __graft_entry__ as g;
g.build()"` or `make -C dgnn_amd/csrc`) importing a symbol raises; calling into it without a GPU
raises from the HIP runtime.  The product never routes through `oracle/` or any CPU path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGNN_LIB_PATH: another build of the same library (kernel variants for A/B measurements, tools/build_variant.sh) -- never a fallback
LIB_PATH = os.environ.get("DGNN_LIB_PATH") or os.path.join(_HERE, "libdgnn_hip.so")

i64, i32, f32, vp = C.c_int64, C.c_int, C.c_float, C.c_void_p

# name -> (restype, argtypes); mirrors include/dgnn_hip.h one to one
SIGNATURES = {
    "dgnn_version": (i32, []),
    "dgnn_last_error_string": (C.c_char_p, []),
    "dgnn_plan_scratch_elems": (i64, [i64, i64]),
    "dgnn_poll_async_error": (i32, []),
    "dgnn_plan_build": (i32, [vp, i64, i64, i64, i64, i64, i32, i32, vp, vp, vp, vp, vp]),
    "dgnn_gather_rows_f32": (i32, [vp, i64, vp, i64, i32, vp, i64, vp]),
    "dgnn_scatter_rows_f32": (i32, [vp, i64, vp, i64, i32, vp, i64, vp]),
    "dgnn_relu": (i32, [vp, i64, vp, vp]),
    "dgnn_relu_bwd": (i32, [vp, vp, i64, vp, vp]),
    "dgnn_sage_aggregate_fwd": (i32, [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64, vp]),
    "dgnn_linear_fwd": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, i32, i64, i32, vp, i64, vp]),
    "dgnn_linear_fwd_x3": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, i32, i64, i32, vp, i64, vp]),
    "dgnn_linear_fwd_x2h": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, i32, i64, i32, vp, i64, vp, vp]),
    "dgnn_linear_fwd_x2h_scratch_elems": (i64, [i64, i32]),
    "dgnn_linear_fwd_x2hp": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, i32, i64, i32, vp, i64, vp, vp]),
    "dgnn_linear_fwd_x2hp_scratch_elems": (i64, [i64, i32, i32, i32]),
    "dgnn_linear_wgrad_x3": (i32, [vp, i64, i32, vp, i64, i32, i64, vp, i64, i32, vp, vp]),
    "dgnn_linear_wgrad_cat_scratch_elems": (i64, [i64, i32, i32, i32]),
    "dgnn_train_set_fused": (i32, [i32]),
    "dgnn_adam_step": (i32, [i32, vp, vp, vp, vp, vp, f32, f32, f32, f32, i64, vp]),
    "dgnn_updated_tail_fwd": (i32, [i64, vp, i64, i32, vp, vp, i32, vp, vp, i32, vp, vp, i32, i32, vp]),
    "dgnn_updated_tail_scratch_elems": (i64, [i64, i32, i32, i32]),
    "dgnn_updated_tail_bwd": (i32, [i64, vp, i64, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "dgnn_updated_stack_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                    i32, i32, vp]),
    "dgnn_updated_stack_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                    vp, vp, vp, vp, vp, i32, i32, vp]),
    "dgnn_linear_wgrad_bf16_cat": (i32, [vp, i32, i64, i32, vp, i64, i32, vp, i64, i32, i32, i64, vp, vp, vp, vp, vp]),
    "dgnn_sage_aggregate_bwd_phi_add": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, i64, vp, i64, vp, i32, vp]),
    "dgnn_linear_fwd_x3_stats": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp]),
    "dgnn_bn_stats_finalize_fold": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, f32, vp, vp, f32, vp, vp, vp]),
    "dgnn_linear_wgrad_x3_cat": (i32, [vp, i64, i32, vp, i64, i32, vp, i64, i32, i64, vp, vp, vp, vp, vp]),
    "dgnn_sage_aggregate_bwd_add": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64, i64, vp, vp, vp, vp]),
    "dgnn_linear_wgrad_scratch_elems": (i64, [i64, i32, i32]),
    "dgnn_linear_wgrad": (i32, [vp, i64, i32, vp, i64, i32, i64, vp, i64, i32, vp, vp]),
    "dgnn_bn_fold": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp]),
    "dgnn_colstats_scratch_elems": (i64, [i64, i32]),
    "dgnn_bn_batch_stats": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, f32, vp, vp]),
    "dgnn_bn_batch_stats_fold": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, f32, vp, vp, f32, vp, vp, vp, vp]),
    "dgnn_scale_shift_act": (i32, [vp, i64, vp, vp, i32, i64, i32, vp, i64, vp]),
    "dgnn_bn_relu_bwd": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, f32, i32, i32, i64, i32, vp, i64, vp, vp, vp, vp]),
    "dgnn_bn_relu_bwd_sums": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, f32, i32, i64, i32, vp, vp, vp]),
    "dgnn_bn_relu_bwd_apply": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, f32, i32, i64, i32, vp, C.c_double, vp, i64, vp]),
    "dgnn_colsum": (i32, [vp, i64, i64, i32, vp, i32, vp, vp]),
    "dgnn_sage_aggregate_bwd_scratch_elems": (i64, [i64, i32, i32]),
    "dgnn_sage_aggregate_bwd": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64,
                                      vp, vp, vp, i64, vp, vp]),
    "dgnn_debug_trace_buffer": (i32, [vp, i64]),
    "dgnn_fill_i32": (i32, [vp, i64, i32, vp]),
    "dgnn_standardize_scratch_doubles": (i64, [i32]),
    "dgnn_standardize_f64": (i32, [vp, i64, i64, i32, i32, vp, i64, vp, vp]),
    "dgnn_cell_centroids_scratch_elems": (i64, [i64]),
    "dgnn_cell_centroids_3dt": (i32, [vp, i64, vp, i64, vp, vp, i64, i64, i64, vp, vp, vp]),
    "dgnn_cell_order_morton_scratch_elems": (i64, [i64]),
    "dgnn_cell_order_morton": (i32, [vp, i64, vp, vp, vp, vp]),
    "dgnn_cell_order_bfs_scratch_elems": (i64, [i64]),
    "dgnn_cell_order_bfs": (i32, [vp, i64, i64, i64, vp, vp, vp, vp]),
    "dgnn_reorder_edges_ref": (i32, [vp, i64, i64, i64, vp, vp, vp, vp, vp]),
    "dgnn_argmax_rows": (i32, [vp, i64, i64, i32, vp, vp]),
    "dgnn_compact_scratch_elems": (i64, [i64]),
    "dgnn_compact_i32": (i32, [vp, vp, i32, i64, vp, vp, vp, vp]),
    "dgnn_interface_flags": (i32, [vp, vp, i64, vp, vp]),
    "dgnn_khop_scratch_elems": (i64, [i64, i64]),
    "dgnn_khop_count": (i32, [vp, vp, i64, i32, vp, vp, vp, vp]),
    "dgnn_khop_expand": (i32, [vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_khop_commit": (i32, [vp, i64, i64, vp, vp, vp]),
    "dgnn_khop_reset": (i32, [vp, i64, vp, vp]),
    "dgnn_khop_blocks_regular": (i32, [vp, vp, vp, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_khop_blocks_regular_start": (vp, [vp, vp, vp, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_khop_blocks_regular_start_rows": (vp, [vp, vp, vp, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                                 i32, vp, vp, vp, vp, vp, vp]),
    "dgnn_khop_blocks_regular_wait": (i32, [vp, i32, vp]),
    "dgnn_decoder_fused_fwd": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, i64, vp]),
    "dgnn_cast_f32_to_bf16": (i32, [vp, i64, i64, i32, i32, vp, i64, vp]),
    "dgnn_rows_unsigned_to_bf16": (i32, [vp, i64, i64, i32, vp, i64, vp]),
    "dgnn_cast_bf16_to_f32": (i32, [vp, i64, i64, i32, vp, i64, vp]),
    "dgnn_sage_layer_fused_fwd_bf16": (i32, [vp, vp, vp, i64, vp, i32, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i64, i32, vp]),
    "dgnn_sage_layer_fused_decoder_fwd_bf16": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32,
                                                     vp, vp, i32, vp, i32, vp]),
    "dgnn_decoder_fused_fwd_bf16": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, i64, i32, vp]),
    "dgnn_sage_aggregate_fwd_bf16": (i32, [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64, vp]),
    "dgnn_sage_aggregate_bwd_bf16": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, i64, vp, i64, vp, i64,
                                           vp, vp, vp, i64, vp, vp]),
    "dgnn_linear_fwd_bf16": (i32, [vp, i64, i32, vp, i64, vp, i64, i32, vp, i64, vp, vp, vp, i32, i64, i32, vp, i64, i32, vp]),
    "dgnn_linear_wgrad_bf16": (i32, [vp, i32, i64, i32, vp, i32, i64, i32, i64, vp, i64, i32, vp, vp]),
    "dgnn_bn_batch_stats_bf16": (i32, [vp, i64, i64, i32, vp, vp, vp, vp, f32, vp, vp]),
    "dgnn_scale_shift_act_bf16": (i32, [vp, i64, vp, vp, i32, i64, i32, vp, i64, vp]),
    "dgnn_bn_relu_bwd_bf16": (i32, [vp, i64, vp, i64, vp, i64, vp, vp, vp, f32, i32, i32, i64, i32, vp, i64, vp, vp, vp, vp]),
    "dgnn_colsum_bf16": (i32, [vp, i64, i64, i32, vp, i32, vp, vp]),
    "dgnn_relu_bf16": (i32, [vp, i64, vp, vp]),
    "dgnn_relu_bwd_bf16": (i32, [vp, vp, i64, vp, vp]),
    "dgnn_sage_layer_fused_fwd": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp,
                                        i64, i32, vp]),
    "dgnn_sage_layer_fused_decoder_fwd": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32,
                                                vp, vp, i32, vp, vp]),
    "dgnn_sage_layer_prepared_bytes": (i64, [i32, i32, i32]),
    "dgnn_sage_layer_prepare": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_sage_layer_fused_fwd_p": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i64, vp, vp]),
    "dgnn_sage_layer_fused_decoder_fwd_p": (i32, [vp, vp, vp, i64, vp, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32,
                                                  vp, vp, i32, vp, vp, vp]),
    "dgnn_static_infer_workspace_bytes": (i64, [i64, i32, vp]),
    "dgnn_static_infer_fwd": (i32, [vp, i64, i64, i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                    vp, vp, i32, i32, i32, vp, vp, vp]),
    "dgnn_static_infer_rings_fwd": (i32, [vp, i64, i64, i64, i32, vp, vp, vp, vp, i32, i64, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp]),
    "dgnn_static_infer_rings_fwd_bf16": (i32, [vp, i64, i64, i64, i32, vp, vp, vp, vp, i32, i64, vp, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp,
                                               vp, vp, vp, vp, i32, vp, vp, i32, i32, vp, vp, vp]),
    "dgnn_static_infer_partitioned_fwd": (i32, [vp, i64, i64, i64, i32, vp, vp, vp, vp, i32, i64, i64, i64, vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp,
                                                vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp]),
    "dgnn_rccl_available": (i32, []),
    "dgnn_comm_unique_id": (i32, [vp]),
    "dgnn_comm_create": (i32, [vp, i32, i32, vp]),
    "dgnn_comm_destroy": (i32, [vp]),
    "dgnn_comm_count": (i32, [vp]),
    "dgnn_wave_specialised_enabled": (i32, []),
    "dgnn_sr_row_bytes": (i64, [i32]),
    "dgnn_sr_pack": (i32, [vp, i64, i32, vp, i64, i32, i64, i32, vp, i64, vp, i32, vp]),
    "dgnn_sr_unpack": (i32, [vp, i64, vp, i32, i32, i32, i64, vp, i64, vp]),
    "dgnn_sr_filter_prepared_bytes": (i64, [i32]),
    "dgnn_sr_prepare_filter": (i32, [vp, vp, i32, vp, vp]),
    "dgnn_sage_aggregate_sr": (i32, [vp, vp, vp, i64, vp, i32, i64, vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_linear_sr": (i32, [vp, i64, vp, i32, vp, i64, vp, i32, vp, vp, vp, vp, vp, i32, i64, i32, vp, i64, vp, vp, i64, vp, vp, i32, vp, vp]),
    "dgnn_halo_plan_create": (i32, [i32, i32, i64, vp, vp, vp, vp]),
    "dgnn_halo_plan_destroy": (i32, [vp]),
    "dgnn_halo_send_rows": (i64, [vp]),
    "dgnn_halo_recv_rows": (i64, [vp]),
    "dgnn_halo_exchange_start": (i32, [vp, vp, vp, i64, i32, i32, vp, vp]),
    "dgnn_halo_exchange_wait": (i32, [vp, vp]),
    "dgnn_edge_chain_fwd": (i32, [vp, i64, i32, vp, i64, vp, i64, i64, vp, i32, vp, i64, vp, vp]),
    "dgnn_edge_chain_fwd_bf16": (i32, [vp, i64, i32, vp, i64, vp, i64, i64, vp, i32, vp, i64, vp, vp]),
    "dgnn_edge_chain_bwd": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i64, vp]),
    "dgnn_edge_chain_bwd_bf16": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp, i64, vp]),
    "dgnn_sage_layer_train_fwd_bf16": (i32, [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, i32,
                                             vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_sage_layer_train_bwd_bf16": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, f32, i32,
                                             vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "dgnn_static_train_fwd": (i32, [i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "dgnn_train_set_aux_stream": (i32, [i32]),
    "dgnn_static_train_scratch_elems": (i64, [i32, vp, vp, vp, i32]),
    "dgnn_static_train_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                    vp, vp, vp, vp, i32, vp]),
    "dgnn_sage_updated_train_scratch_elems": (i64, [i64, i64, i32, i32, i32]),
    "dgnn_sage_updated_train_fwd": (i32, [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, i32, vp]),
    "dgnn_sage_updated_train_bwd": (i32, [vp, vp, vp, vp, i64, i64, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp,
                                          vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "dgnn_kl_cell_loss_scratch_doubles": (i64, [i64]),
    "dgnn_kl_cell_loss_fwd": (i32, [vp, i64, vp, i64, vp, i64, i32, i64, vp, vp, vp, vp]),
    "dgnn_kl_cell_loss_bwd": (i32, [vp, i64, vp, i64, vp, i64, i32, i64, vp, vp, vp, i64, vp]),
    "dgnn_kl_cell_loss_step": (i32, [vp, i64, vp, i64, vp, i64, i32, i64, vp, vp, vp, vp, vp, i64, vp]),
    "dgnn_sage_layer_train_scratch_elems": (i64, [i64, i64, i32, i32, i32]),
    "dgnn_sage_layer_train_fwd": (i32, [vp, vp, vp, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, i32,
                                        vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "dgnn_sage_layer_train_bwd": (i32, [vp, vp, vp, vp, i64, i64, vp, i64, i32, vp, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp, f32, i32,
                                        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
}

_lib = None


class DgnnError(RuntimeError):
    pass


def lib():
    """Loads libdgnn_hip.so once and attaches argtypes.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DgnnError("%s is missing: build it with `make -C dgnn_amd/csrc` (hipcc, gfx950); "
                            "dgnn_amd has no CPU fallback" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


_calls = 0


def check(code: int, what: str = "", poll: bool = False):
    """Raises DgnnError for a non-zero status.  Asynchronous kernel errors (an edge_index entry out of range, see
    dgnn_poll_async_error in the header) are polled -- a host memory read, no sync -- at the entry points that consume index
    data (`poll=True`: plan and block builders) and at every 16th call of any entry point, so they surface within a few
    launches without a second library call per launch."""
    global _calls
    _calls += 1
    if code == 0 and (poll or (_calls & 15) == 0):
        code = lib().dgnn_poll_async_error()
        if code != 0:
            what = "an earlier dgnn kernel (reported at %s)" % (what or "this call")
    if code != 0:
        msg = lib().dgnn_last_error_string().decode()
        raise DgnnError("%s failed (%d): %s" % (what or "dgnn call", code, msg))


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    """torch's current stream of the CURRENT device; callers launch under `on_device_of` so that this is the device
    the tensors live on (the reference addresses its GPU as "cuda:<n>" and never calls set_device: run.py:98,127).
    (torch.cuda.current_stream() builds a Stream object through three Python layers, ~10 us; the training step asks ~30 times.)"""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device_of(fn):
    """Decorator for functions that launch kernels: runs `fn` with the device of its first GPU tensor argument current,
    so kernels, torch's stream and every allocation inside agree on the device even when it is not device 0 / not the
    thread's current device.  Tensors on different GPUs in one call raise."""
    import functools

    import torch

    _current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device

    def _tensors(a, k):
        for t in a:
            if isinstance(t, torch.Tensor):
                yield t
        for t in k.values():
            if isinstance(t, torch.Tensor):
                yield t

    @functools.wraps(fn)
    def wrapper(*a, **k):
        dev = None
        for t in _tensors(a, k):
            if t.is_cuda:
                if dev is None:
                    dev = t.device
                elif t.device != dev:
                    raise DgnnError("%s: tensors on different devices (%s, %s)" % (fn.__name__, dev, t.device))
        if dev is not None and dev.index != _current_device():
            with torch.cuda.device(dev):
                return fn(*a, **k)
        return fn(*a, **k)

    return wrapper
