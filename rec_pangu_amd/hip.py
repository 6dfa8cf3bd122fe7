"""ctypes binding of librecpangu_hip.so (C ABI: include/rec_pangu_hip.h).

This is the only place Python touches the kernels.  PyTorch is plumbing here: it owns device
memory (`tensor.data_ptr()`), and the current HIP stream (`torch.cuda.current_stream()`); no
torch type crosses the ABI.  There is NO fallback: if the shared object is missing or a call
fails, a RuntimeError is raised — a CUDA(HIP)-resident tensor never silently takes another path.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 106  # csrc/common.hip: rp_version() — bumped with every change of the entry points' prototypes
LIB_PATH = os.environ.get("RP_LIB_PATH") or os.path.join(_HERE, "lib", "librecpangu_hip.so")  # (override: A/B builds)
MAX_FIELDS = 64

ACT_NONE, ACT_RELU, ACT_MASK, ACT_TANH, ACT_SIGMOID, ACT_LEAKY = 0, 1, 2, 3, 4, 5

_lib = None

_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t
_f64 = C.c_double
_SIGNATURES = {
    "rp_version": (C.c_int, []),
    "rp_last_error": (C.c_char_p, []),
    "rp_launch_count": (C.c_uint64, []),
    "rp_embed_gather_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "rp_embed_gather_linear_fits": (C.c_int, [_i32, _i32, _i32, _i64, _i64]),
    "rp_embed_gather_linear_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp, _vp]),
    "rp_embed_gather_linear_fwd_bf16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp,
                                                  _i64, _vp, _vp, _vp, _vp, _vp]),
    "rp_linear_wgrad_gather_fits": (C.c_int, [_i64, _i32, _i32, _i32]),
    "rp_linear_wgrad_gather": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _sz,
                                         _vp]),
    "rp_sort_workspace_bytes": (C.c_int, [_i64, C.POINTER(_sz)]),
    "rp_sort_pairs_i32": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _i64, _i32, _vp]),
    "rp_sort_pairs_fields_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_sort_pairs_fields_i32": (C.c_int, [_vp, _sz, _vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp]),
    "rp_embed_grad_reduce_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_embed_grad_reduce": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "rp_embed_grad_gemm_fits": (C.c_int, [_i32, _i32, _i64, _i64]),
    "rp_embed_grad_gemm": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32,
                                     C.c_uint64, _vp, _sz, _vp]),
    "rp_embed_grad_tiny_workspace_bytes": (C.c_int, [_i64, C.POINTER(_sz)]),
    "rp_embed_grad_tiny": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i64,
                                     _vp, _sz, _vp]),
    "rp_embed_grad_ss_workspace_bytes": (C.c_int, [_i64, _i64, _i32, C.c_uint64, C.POINTER(_sz)]),
    "rp_embed_grad_ss_mark_sizes": (C.c_int, [_i64, _i32, C.POINTER(_sz), C.POINTER(_sz)]),
    "rp_embed_grad_ss_mark": (C.c_int, [_vp, _i64, _i64, C.c_uint64, _vp, _vp, _vp, _vp]),
    "rp_embed_grad_ss": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, C.c_uint64, _vp,
                                   _vp, _i64, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "rp_embed_grad_smp_fits": (C.c_int, [_i32, _i32, _i64]),
    "rp_embed_grad_smp_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_embed_grad_smp_mark_scratch": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_embed_grad_smp_mark": (C.c_int, [_vp, _vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp, _vp]),
    "rp_embed_grad_smp": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32,
                                    _vp, _i64, _i32, _vp, _sz, _vp]),
    "rp_embed_grad_reduce_rows": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _vp, _sz, _vp]),
    "rp_embed_grad_seg_fits": (C.c_int, [_i32, _i32, _i64]),
    "rp_embed_grad_seg_workspace_bytes": (C.c_int, [_i64, _i64, _i32, C.POINTER(_sz)]),
    "rp_embed_grad_seg": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, C.c_uint64, _vp,
                                    _vp, _i64, _vp, _sz, _vp]),
    "rp_zero_rows": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "rp_linear_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "rp_linear_fwd_rowadd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _i64, _i32, _vp]),
    "rp_linear_wgrad_workspace_bytes": (C.c_int, [_i64, _i32, _i32, C.POINTER(_sz)]),
    "rp_linear_wgrad": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "rp_transpose": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "rp_copy_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "rp_add_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    "rp_transpose_copy": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _i64, _vp]),
    "rp_pieces_ld": (C.c_int64, [_i32, _i32]),
    "rp_pieces_pack": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp]),
    "rp_linear_fwd_pieces": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "rp_plan_begin": (C.c_int, [C.POINTER(_vp)]),
    "rp_plan_section": (C.c_int, [_i32]),
    "rp_plan_fork_here": (C.c_int, []),
    "rp_plan_join": (C.c_int, []),
    "rp_plan_fork2_mark": (C.c_int, []),
    "rp_stream_create_low": (C.c_int, [C.POINTER(_vp)]),
    "rp_plan_is_recording": (C.c_int, []),
    "rp_plan_end": (C.c_int, [_vp]),
    "rp_plan_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "rp_plan_replay": (C.c_int, [_vp, _vp]),
    "rp_plan_set_streams": (C.c_int, [_vp, _vp, _vp]),
    "rp_plan_set_probe": (C.c_int, [_vp, _i32]),
    "rp_plan_probe_ms": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "rp_marker_create": (C.c_int, [C.POINTER(_vp)]),
    "rp_marker_record": (C.c_int, [_vp, _vp]),
    "rp_marker_wait": (C.c_int, [_vp]),
    "rp_marker_destroy": (C.c_int, [_vp]),
    "rp_plan_slowest_call": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]),
    "rp_plan_launch_name": (C.c_int, [_vp, _i32, C.c_char_p, _i32, C.POINTER(_i32)]),
    "rp_plan_join_side": (C.c_int, []),
    "rp_plan_host_mark": (C.c_int, [C.POINTER(_i32)]),
    "rp_plan_host_marks": (C.c_int, [_vp, C.POINTER(_i32)]),
    "rp_plan_replay_segment": (C.c_int, [_vp, _i32, _vp]),
    "rp_fill_words": (C.c_int, [_vp, _i64, C.c_uint32, _vp]),
    "rp_plan_side2_sync": (C.c_int, []),
    "rp_plan_bind_inputs": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32)]),
    "rp_plan_bind_report": (C.c_int, [_vp, _vp, _vp, _i32, _vp, C.POINTER(_i32)]),
    "rp_plan_set_inputs": (C.c_int, [_vp, _vp, _i32]),
    "rp_plan_inline_count": (C.c_int, [_vp, C.POINTER(_i32)]),
    "rp_plan_destroy": (C.c_int, [_vp]),
    "rp_graph_node_counts": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "rp_multi_copy": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "rp_relu_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "rp_act_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "rp_act_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "rp_crossnet_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "rp_crossnet_bwd_rows": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp]),
    "rp_crossnet_param_grads": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rp_cin_layer_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i64, _vp]),
    "rp_cin_layer_bwd_x": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i32,
                                     _i32, _i32, _i64, _vp]),
    "rp_cin_layer_bwd_w_workspace_bytes": (C.c_int, [_i64, _i32, _i32, _i32, C.POINTER(_sz)]),
    "rp_cin_layer_bwd_w": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i64, _vp,
                                     _sz, _vp]),
    "rp_field_attention_fits": (C.c_int, [_i32, _i32, _i32, _i32, _i32]),
    "rp_field_attention_fwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _i64, _vp]),
    "rp_field_attention_bwd_workspace_bytes": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _i32, C.POINTER(_sz)]),
    "rp_field_attention_bwd": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp, _i64,
                                         _vp, _sz, _vp]),
    "rp_mmoe_combine_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "rp_mmoe_combine_bwd": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp]),
    "rp_cin_bs_fits": (C.c_int, [_i32, _i32, _i32]),
    "rp_cin_bs_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "rp_cin_bs_bwd_x": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _i64, _vp]),
    "rp_cin_bs_bwd_w_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_cin_bs_bwd_w": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "rp_cin_pair_fits": (C.c_int, [_i32, _i32, _i32]),
    "rp_cin_pair_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp]),
    "rp_cin_pair_bwd_x": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _i64, _i32, _vp]),
    "rp_cin_pair_bwd_w_workspace_bytes": (C.c_int, [_i64, _i32, _i32, C.POINTER(_sz)]),
    "rp_cin_pair_bwd_w": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _sz, _vp]),
    "rp_cin_last_fits": (C.c_int, [_i32, _i32, _i32]),
    "rp_cin_last_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp]),
    "rp_cin_last_bwd_x": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _i64, _i64, _vp]),
    "rp_cin_last_bwd_v_workspace_bytes": (C.c_int, [_i64, _i32, _i32, C.POINTER(_sz)]),
    "rp_cin_last_bwd_v": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _vp, _i64, _vp, _sz, _vp]),
    "rp_cin_pair_pieces": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "rp_cin_head_params_fwd": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "rp_cin_head_params_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rp_add_scalars": (C.c_int, [_vp, _i64, _vp, C.c_float, _vp, _vp]),
    "rp_sum_all": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rp_attention_core_fits": (C.c_int, [_i32, _i32, _i32]),
    "rp_attention_core_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _i64, _vp]),
    "rp_attention_core_bwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _i64, _vp, _i64, _i64,
                                        _vp]),
    "rp_set_matmul_precision": (C.c_int, [_i32]),
    "rp_get_matmul_precision": (C.c_int, []),
    "rp_fm_pool_fwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _i64, _vp]),
    "rp_fm_pool_bwd": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp]),
    "rp_batchnorm_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_batchnorm_train_fwd": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "rp_batchnorm_train_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _vp, _sz,
                                         _vp]),
    "rp_batchnorm_apply": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rp_batchnorm_apply_bwd": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rp_batchnorm_update_running": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _i64, _i32, _vp]),
    "rp_batchnorm_colsum": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "rp_batchnorm_bwd_sums": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _sz, _vp]),
    "rp_batchnorm_bwd_apply": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rp_dice_gate_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _vp]),
    "rp_dice_gate_bwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp]),
    "rp_mlp_tail_fits": (C.c_int, [_i32, _i32, _i64]),
    "rp_mlp_tail_fwd": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "rp_mlp_tail_bwd_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_mlp_tail_bwd": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _sz, _vp]),
    "rp_mlp_tail_bwd_parts": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _sz, _i32, _vp]),
    "rp_mlp_tail_loss_partials": (C.c_int, [_i64]),
    "rp_mlp_tail_fwd_bce": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _f32, _vp, _vp, _i64, _vp]),
    "rp_mlp_tail_bwd_bce": (C.c_int, [_vp, _vp, _vp, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp,
                                      _sz, _i32, _vp]),
    "rp_loss_finish": (C.c_int, [_vp, _i32, _f32, _vp, _vp]),
    "rp_dropout_fwd": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, C.c_uint64, C.c_uint64, _vp]),
    "rp_dropout_fwd_dev": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, C.c_uint64, C.c_uint64, _vp, _vp]),
    "rp_counter_add_u64": (C.c_int, [_vp, C.c_uint64, _vp]),
    "rp_dropout_bwd": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _i32, _f32, _vp]),
    "rp_loss_partials": (C.c_int, [_i64]),
    "rp_sigmoid_bce_fwd": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp]),
    "rp_sigmoid_bce_fwd_accum": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _f32, _f32, _vp, _vp, _vp, _vp]),
    "rp_sigmoid_bce_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _i32, _vp, _vp]),
    "rp_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _f64, _f64, _f64, _f64, _i64, _i32, _vp, _vp, _vp]),
    "rp_counter_add": (C.c_int, [_vp, _i32, _vp]),
    "rp_counters_add": (C.c_int, [_vp, _i32, _i32, _vp]),
    "rp_accumulate": (C.c_int, [_vp, _vp, _i64, _vp]),
    "rp_embed_keys": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "rp_shard_keys": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp]),
    "rp_route_workspace_bytes": (C.c_int, [_i64, _i32, C.POINTER(_sz)]),
    "rp_route_build": (C.c_int, [_vp, _sz, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rp_route_pad": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rp_route_field_major": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rp_adam_step_scalars": (C.c_int, [_f64, _f64, _f64, _f64, _i64, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rp_adam_step_scalars_range": (C.c_int, [_f64, _f64, _f64, _f64, _i64, _i64, _vp]),
    "rp_lazy_adam_rows": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _f64, _f64, _f64,
                                    _vp, _i64, _vp, _vp]),
    "rp_lazy_adam_flush": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _vp, _i64, _vp]),
    "rp_lazy_adam_catchup": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64,
                                       _vp, _i64, _vp, _vp, _vp]),
    "rp_lazy_adam_flush_deferred": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _vp, _i64,
                                              _vp, _vp]),
    "rp_rows_to_bf16": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "rp_linear_wgrad_xbf16": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "rp_embed_gather_pool_fwd": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "rp_embed_pool_bwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i32, _vp, _sz, _vp]),
    "rp_seq_pool_fwd": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "rp_seq_pool_bwd": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "rp_lazy_adam_cf_terms": (C.c_int, [_f64, C.POINTER(C.c_int)]),
    "rp_lazy_adam_cf_table": (C.c_int, [_vp, _i64, _i64, _f64, _f64, _vp, _i64, _i64, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES.keys())


def lib():
    """Load the shared object once; raise loudly if it is not there (no CPU stand-in exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C rec_pangu_amd/csrc` "
                "(or __graft_entry__.build()). The HIP path has no fallback.")
        handle = C.CDLL(LIB_PATH)
        handle.rp_version.restype = C.c_int
        if handle.rp_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} is version {handle.rp_version()} of the C ABI, these bindings are for {ABI_VERSION}: "
                               "rebuild it (`make -C rec_pangu_amd/csrc`, or __graft_entry__.build())")
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
        mode = os.environ.get("RP_MATMUL_PRECISION")  # fp32 | bf16 | bf16x3 | bf16x6 | auto (library default)
        if mode:
            if handle.rp_set_matmul_precision(MATMUL_MODES[mode]) != 0:
                raise RuntimeError(f"RP_MATMUL_PRECISION={mode}: rejected by the library")
    return _lib


def available() -> bool:
    return os.path.exists(LIB_PATH)


def launch_count() -> int:
    return int(lib().rp_launch_count())


# ---- HIP-resident modules that take a torch (ATen) path: counted and warned about once, never silent ----------------
_torch_paths: dict = {}


def note_torch_path(what: str) -> None:
    """A module whose tensors live on the HIP device is about to compose its result from torch device ops instead of the
    library's kernels (a configuration no kernel covers: a masked / cross attention, a CIN over more than 32 fields, a
    non-ReLU activation module inside an MLP ...).  Results are the same; the launch is not ours.  Counted per reason
    (torch_path_count / torch_paths) and warned about once per reason; RP_STRICT_HIP=1 turns it into an error."""
    n = _torch_paths.get(what, 0)
    _torch_paths[what] = n + 1
    if os.environ.get("RP_STRICT_HIP", "0") == "1":
        raise RuntimeError(f"RP_STRICT_HIP=1: a HIP-resident module took a torch path: {what}")
    if n == 0:
        import warnings
        warnings.warn(f"rec_pangu_amd: HIP-resident module on a torch (ATen) path, not on the library's kernels: {what} "
                      "(reported once; hip.torch_paths() has the counts)", RuntimeWarning, stacklevel=3)


def torch_path_count() -> int:
    return sum(_torch_paths.values())


def torch_paths() -> dict:
    return dict(_torch_paths)


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().rp_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def _stream() -> int:
    # raw handle of torch's current stream: torch.cuda.current_stream() builds a Stream object (~10 us per call,
    # dozens of launches per step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"expected a HIP-device tensor, got one on {t.device}")
    return t.data_ptr()


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a HIP-device tensor, got {t.device}")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")


def _rowmajor(t: torch.Tensor, name: str) -> int:
    """2-D tensor with unit inner stride -> leading dimension."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError(f"{name}: need a 2-D tensor with unit inner stride, got {tuple(t.shape)}/{t.stride()}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


# ---- optional per-entry-point timing with HIP events on the launch stream (bench.py's roofline) ----
_timing = None


def enable_timing(on: bool = True, only=None):
    """Record a HIP-event pair around every C-ABI call (`only`: just the named entry points).  An event record is
    a barrier packet in the queue: timing every launch of a step serialises kernels that would otherwise overlap
    and costs host time, so a throughput measurement should restrict it to the few launches it reports."""
    global _timing, _timing_only
    _timing = {} if on else None
    _timing_only = None if only is None else frozenset(only)


_timing_only = None
_timing_paused = False


def pause_timing(paused: bool):
    """Keep the recorded events but stop/resume recording new ones (lets a benchmark sample every n-th step)."""
    global _timing_paused
    _timing_paused = bool(paused)


class _Timed:
    """`tag`: a launch-shape suffix (bench.py then reports one row per kernel shape instead of one per entry point);
    `nbytes` / `flops`: algorithmic bytes moved / flops of this launch where the entry point knows them itself."""
    __slots__ = ("name", "key", "ev", "meta")

    def __init__(self, name, tag=None, nbytes=0, flops=0):
        self.name = name
        self.key = name if tag is None else f"{name}[{tag}]"
        self.meta = (nbytes, flops)

    def __enter__(self):
        self.ev = None
        if _timing is not None and not _timing_paused and (_timing_only is None or self.name in _timing_only):
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, *exc):
        if self.ev is not None:
            self.ev[1].record()
            _timing.setdefault(self.key, []).append(self.ev)
            if self.meta != (0, 0):
                _timing_meta[self.key] = self.meta
        return False


_timing_meta = {}


def timing_summary():
    """{entry point[shape tag]: (calls, mean milliseconds)} for the calls recorded since enable_timing(True)."""
    out = {}
    for name, evs in (_timing or {}).items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[name] = (len(ms), sum(ms) / max(len(ms), 1))
    return out


def timing_meta():
    """{entry point[shape tag]: (algorithmic bytes, flops) per launch} for the rows that carry them (the GEMMs)."""
    return dict(_timing_meta)


_PTR_ARRAYS: dict = {}  # tuple of device addresses -> the ctypes array holding them (the same lists recur every step)


def _ptr_array(tensors: Sequence[torch.Tensor]):
    """HOST array of the tensors' device addresses for a `*_ptrs` argument (copied into the launch packet by the
    library).  Built once per distinct address tuple: the element-wise fill of a ctypes array was the largest single item
    of the host's time per step (10 calls x up to 39 tensors: ~0.25 ms of a ~1.0 ms enqueue, host_profile.py)."""
    ptrs = tuple([t.data_ptr() for t in tensors])
    arr = _PTR_ARRAYS.get(ptrs)
    if arr is None:
        for i, t in enumerate(tensors):
            if not t.is_cuda:  # a host pointer inside a launch packet is a GPU memory fault, not an exception
                raise RuntimeError(f"Expected all tensors to be on the same device: input {i} is on {t.device} but the "
                                   "model is on the HIP device (move the batch with .to(device))")
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        if len(_PTR_ARRAYS) >= 512:
            _PTR_ARRAYS.clear()
        _PTR_ARRAYS[ptrs] = arr
    return arr


# ----------------------------------------------------------------------------------------------
def embed_gather_fwd(arena, row_base, row_count, idx: List[torch.Tensor], dense: List[torch.Tensor], ldx: int,
                     want_fm: bool, want_sum: bool, want_keys: bool, err_flag: torch.Tensor):
    _req(arena, torch.float32, "arena")
    F, ND = len(idx), len(dense)
    B, D = idx[0].shape[0], arena.shape[1]
    for t in idx:
        _req(t, torch.int64, "index")
        if t.dim() != 1 or t.shape[0] != B or not t.is_contiguous():
            raise RuntimeError("index tensors must be contiguous int64 [B]")
    for t in dense:
        _req(t, torch.float32, "dense")
        if t.dim() != 1 or t.shape[0] != B or not t.is_contiguous():
            raise RuntimeError("dense tensors must be contiguous float32 [B]")
    dev = arena.device
    x = torch.empty((B, ldx), dtype=torch.float32, device=dev)
    fm = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_fm else None
    ssum = torch.empty((B, D), dtype=torch.float32, device=dev) if want_sum else None
    keys = torch.empty((F * B,), dtype=torch.int32, device=dev) if want_keys else None
    with _Timed("embed_gather_fwd", f"D={D}"):
        _check(lib().rp_embed_gather_fwd(arena.data_ptr(), row_base.data_ptr(), row_count.data_ptr(), _ptr_array(idx), F,
                                     _ptr_array(dense), ND, B, D, x.data_ptr(), ldx, _ptr(fm), _ptr(ssum), _ptr(keys),
                                     err_flag.data_ptr(), _stream()), "rp_embed_gather_fwd")
    return x, fm, ssum, keys


def embed_gather_linear_fits(D: int, F: int, ND: int, hidden: int, ldx: int, W) -> bool:
    return F <= 32 and bool(lib().rp_embed_gather_linear_fits(D, ND, hidden, ldx, _rowmajor(W, "W"))) and W.data_ptr() % 16 == 0


def embed_gather_linear_fwd(arena, row_base, row_count, idx: List[torch.Tensor], dense: List[torch.Tensor], ldx: int, W, bias,
                            want_fm: bool, want_sum: bool, want_keys: bool, err_flag: torch.Tensor, x_mode: str = "full"):
    """the gather fused with the 64-wide Linear + ReLU that consumes it -> (x, h1 [B, 64], fm, ssum, keys).
    x_mode: "full" = x [B, ldx] is stored; "dense" = only the dense columns, as xd [B, 64] (the weight gradient gathers the
    embedding rows itself: linear_wgrad_gather); "none" = nothing is stored (inference)."""
    _req(arena, torch.float32, "arena")
    _req(W, torch.float32, "W")
    F, ND = len(idx), len(dense)
    B, D = idx[0].shape[0], arena.shape[1]
    for t in idx:
        _req(t, torch.int64, "index")
        if t.dim() != 1 or t.shape[0] != B or not t.is_contiguous():
            raise RuntimeError("index tensors must be contiguous int64 [B]")
    for t in dense:
        _req(t, torch.float32, "dense")
        if t.dim() != 1 or t.shape[0] != B or not t.is_contiguous():
            raise RuntimeError("dense tensors must be contiguous float32 [B]")
    dev = arena.device
    assert x_mode in ("full", "dense", "none")
    x = torch.empty((B, ldx), dtype=torch.float32, device=dev) if x_mode == "full" else None
    xd = torch.empty((B, 64), dtype=torch.float32, device=dev) if x_mode == "dense" else None
    h1 = torch.empty((B, 64), dtype=torch.float32, device=dev)
    fm = torch.empty((B, 1), dtype=torch.float32, device=dev) if want_fm else None
    ssum = torch.empty((B, D), dtype=torch.float32, device=dev) if want_sum else None
    keys = torch.empty((F * B,), dtype=torch.int32, device=dev) if want_keys else None
    K = F * D + ND
    # algorithmic bytes (rows + ids read, the [B, F*D+ND] output written — counted whether or not the launch
    # stores it, the figure the gather is priced on) + h1
    with _Timed("embed_gather_linear_fwd", f"D={D}", B * (F * (D * 4 + 8) + (F * D + ND) * 4 + 64 * 4), 2 * B * K * 64):
        _check(lib().rp_embed_gather_linear_fwd(arena.data_ptr(), row_base.data_ptr(), row_count.data_ptr(), _ptr_array(idx), F,
                                                _ptr_array(dense), ND, B, D, _ptr(x), ldx, W.data_ptr(), _rowmajor(W, "W"),
                                                _ptr(bias), h1.data_ptr(), _ptr(fm), _ptr(ssum), _ptr(keys),
                                                err_flag.data_ptr(), _ptr(xd), _stream()), "rp_embed_gather_linear_fwd")
    return (x if x_mode == "full" else xd), h1, fm, ssum, keys


def embed_gather_linear_fwd_bf16(arena_bf16, row_base, row_count, idx: List[torch.Tensor], dense: List[torch.Tensor], W, bias,
                                 err_flag: torch.Tensor, train_ldx: int = 0, want_keys: bool = False, dense_only: bool = False):
    """bf16-storage: the fused lookup + FM + first Linear over a bf16 copy of the arena -> (h1 [B, 64], fm [B, 1]);
    train_ldx > 0 (the bf16-storage TRAINING mode): -> (x_bf16 [B, train_ldx] bfloat16, h1, fm, ssum [B, 64], keys or None);
    dense_only (round 5, with train_ldx > 0): no activation is stored — x is xd [B, 64] fp32, the dense columns"""
    _req(arena_bf16, torch.bfloat16, "arena_bf16")
    _req(W, torch.float32, "W")
    F, ND = len(idx), len(dense)
    B, D = idx[0].shape[0], arena_bf16.shape[1]
    dev = arena_bf16.device
    h1 = torch.empty((B, 64), dtype=torch.float32, device=dev)
    fm = torch.empty((B, 1), dtype=torch.float32, device=dev)
    K = F * D + ND
    if train_ldx:
        x16 = None if dense_only else torch.empty((B, train_ldx), dtype=torch.bfloat16, device=dev)
        xd = torch.empty((B, 64), dtype=torch.float32, device=dev) if dense_only else None
        ssum = torch.empty((B, D), dtype=torch.float32, device=dev)
        keys = torch.empty((F * B,), dtype=torch.int32, device=dev) if want_keys else None
        # algorithmic bytes: bf16 rows + ids read, the bf16 activation "written" (SURVEY 8d's figure, whether or not the launch
        # stores it), h1 + the field sums
        with _Timed("embed_gather_linear_fwd_bf16", f"D={D}", B * (F * (D * 2 + 8) + (F * D + ND) * 2 + 64 * 4 + D * 4), 2 * B * K * 64):
            _check(lib().rp_embed_gather_linear_fwd_bf16(arena_bf16.data_ptr(), row_base.data_ptr(), row_count.data_ptr(),
                                                         _ptr_array(idx), F, _ptr_array(dense), ND, B, D, W.data_ptr(),
                                                         _rowmajor(W, "W"), _ptr(bias), h1.data_ptr(), fm.data_ptr(),
                                                         _ptr(x16), train_ldx, ssum.data_ptr(), _ptr(keys),
                                                         err_flag.data_ptr(), _ptr(xd), _stream()), "rp_embed_gather_linear_fwd_bf16")
        return (xd if dense_only else x16), h1, fm, ssum, keys
    # algorithmic bytes: SURVEY 8d's bf16 figure — bf16 rows + int64 ids read, the [B, F*D] bf16 output "written" (it is
    # consumed in registers here) = F * (2 D + 8) + 2 F D per sample (6 864 B at Criteo shape) — plus h1
    with _Timed("embed_gather_linear_fwd_bf16", f"D={D}", B * (F * (D * 2 + 8) + F * D * 2 + 64 * 4), 2 * B * K * 64):
        _check(lib().rp_embed_gather_linear_fwd_bf16(arena_bf16.data_ptr(), row_base.data_ptr(), row_count.data_ptr(),
                                                     _ptr_array(idx), F, _ptr_array(dense), ND, B, D, W.data_ptr(),
                                                     _rowmajor(W, "W"), _ptr(bias), h1.data_ptr(), fm.data_ptr(),
                                                     None, 0, None, None, err_flag.data_ptr(), None, _stream()),
               "rp_embed_gather_linear_fwd_bf16")
    return h1, fm


def sort_workspace(n: int, device) -> torch.Tensor:
    """a workspace rp_sort_pairs_i32 accepts for n pairs (callers that must not allocate per call keep one)"""
    nbytes = _sz(0)
    _check(lib().rp_sort_workspace_bytes(n, C.byref(nbytes)), "rp_sort_workspace_bytes")
    return torch.empty((nbytes.value,), dtype=torch.uint8, device=device)


def sort_pairs(keys: torch.Tensor, end_bit: int = 32, out=None, workspace=None):
    """out = (sorted keys, positions) to write into (persistent buffers of the captured-step path), else fresh tensors;
    workspace: a persistent sort_workspace(n) (else one is allocated per call)"""
    _req(keys, torch.int32, "keys")
    n = keys.numel()
    nbytes = _sz(0)
    _check(lib().rp_sort_workspace_bytes(n, C.byref(nbytes)), "rp_sort_workspace_bytes")
    if workspace is not None:
        assert workspace.numel() >= nbytes.value and workspace.device == keys.device
        ws = workspace
    else:
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=keys.device)
    ko, po = out if out is not None else (torch.empty_like(keys), torch.empty_like(keys))
    assert ko.shape == keys.shape and po.shape == keys.shape and ko.dtype == po.dtype == torch.int32
    _held(keys, ws, ko, po)
    with _Timed("sort_pairs_i32"):
        _check(lib().rp_sort_pairs_i32(ws.data_ptr(), nbytes.value, keys.data_ptr(), ko.data_ptr(), po.data_ptr(), n,
                                   end_bit, _stream()), "rp_sort_pairs_i32")
    return ko, po


_SORT_FIELDS: dict = {}


def sort_fields_workspace(B: int, F: int, device) -> torch.Tensor:
    nbytes = _sz(0)
    _check(lib().rp_sort_pairs_fields_workspace_bytes(B, F, C.byref(nbytes)), "rp_sort_pairs_fields_workspace_bytes")
    return torch.empty((nbytes.value,), dtype=torch.uint8, device=device)


def sort_pairs_fields(keys: torch.Tensor, B: int, field_rows, out=None, workspace=None):
    """rp_sort_pairs_fields_i32: the pair list of one lookup (keys[f * B + b] = arena row, the tables back to back in field
    order with `field_rows` rows each) sorted field segment by field segment — the result of sort_pairs(keys, bits of the arena),
    bit for bit, in 49 instead of 78 field-passes at Criteo shape.  out / workspace as sort_pairs."""
    _req(keys, torch.int32, "keys")
    F = len(field_rows)
    assert keys.numel() == F * B
    ck = tuple(int(r) for r in field_rows)
    arrs = _SORT_FIELDS.get(ck)
    if arrs is None:
        if len(_SORT_FIELDS) > 64:
            _SORT_FIELDS.clear()
        base, acc = [], 0
        for r in ck:
            base.append(acc)
            acc += r
        arrs = _SORT_FIELDS[ck] = ((C.c_int64 * F)(*base), (C.c_int64 * F)(*ck))
    nbytes = _sz(0)
    _check(lib().rp_sort_pairs_fields_workspace_bytes(B, F, C.byref(nbytes)), "rp_sort_pairs_fields_workspace_bytes")
    if workspace is not None and workspace.numel() >= nbytes.value and workspace.device == keys.device:
        ws = workspace
    else:
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=keys.device)
    ko, po = out if out is not None else (torch.empty_like(keys), torch.empty_like(keys))
    assert ko.shape == keys.shape and po.shape == keys.shape and ko.dtype == po.dtype == torch.int32
    _held(keys, ws, ko, po)
    with _Timed("sort_pairs_i32"):  # (the same row of the per-kernel table as the plain sort: it is that sort)
        _check(lib().rp_sort_pairs_fields_i32(ws.data_ptr(), ws.numel(), keys.data_ptr(), ko.data_ptr(), po.data_ptr(), B, F,
                                              arrs[0], arrs[1], _stream()), "rp_sort_pairs_fields_i32")
    return ko, po


def embed_grad_reduce(sorted_keys, sorted_pos, B: int, D: int, dx, gfm, sum_in, arena, grad_arena, accumulate: bool):
    _req(grad_arena, torch.float32, "grad_arena")
    ldx = 0
    if dx is not None:
        _req(dx, torch.float32, "dx")
        ldx = _rowmajor(dx, "dx")
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_reduce_workspace_bytes(sorted_keys.numel(), D, C.byref(nbytes)),
           "rp_embed_grad_reduce_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    with _Timed("embed_grad_reduce", f"D={D}"):
        _check(lib().rp_embed_grad_reduce(sorted_keys.data_ptr(), sorted_pos.data_ptr(), sorted_keys.numel(), B, D,
                                      _ptr(dx), ldx, _ptr(gfm), _ptr(sum_in), _ptr(arena), grad_arena.data_ptr(),
                                      int(accumulate), ws.data_ptr(), nbytes.value, _stream()), "rp_embed_grad_reduce")


def embed_grad_gemm_fits(D: int, hidden: int, dh, wt) -> bool:
    return bool(lib().rp_embed_grad_gemm_fits(D, hidden, _rowmajor(dh, "dh"), _rowmajor(wt, "wt"))) \
        and dh.data_ptr() % 16 == 0 and wt.data_ptr() % 16 == 0


def embed_grad_tiny(keys, B: int, tiny, dh, wt, gfm, sum_in, arena, grad_arena, accumulate: bool, keep=None, dw=None):
    """rp_embed_grad_tiny: the gradient rows of the tiny tables `tiny` = [(field, first arena row, rows), ...] from the
    unsorted pair keys [F * B] (sample-major one-hot GEMMs).  keep: a list that receives the workspace (launches that run
    beside later ones inside a recorded plan).  dw [64, K]: the tiny tables' columns of the first layer's weight gradient
    are written into it as well (the companion of embed_grad_seg)."""
    _req(keys, torch.int32, "keys")
    _req(dh, torch.float32, "dh")
    n = len(tiny)
    ckey = tuple(tiny)
    arrs = _TINY_ARRAYS.get(ckey)
    if arrs is None:
        arrs = _TINY_ARRAYS[ckey] = ((C.c_int32 * n)(*[t[0] for t in tiny]), (C.c_int64 * n)(*[t[1] for t in tiny]),
                                     (C.c_int32 * n)(*[t[2] for t in tiny]))
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_tiny_workspace_bytes(B, C.byref(nbytes)), "rp_embed_grad_tiny_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    if keep is not None:
        keep.append(ws)
    with _Timed("embed_grad_tiny", f"{n} tables", B * (64 * 8 + 4 + 4 * n)):
        _check(lib().rp_embed_grad_tiny(keys.data_ptr(), B, arrs[0], arrs[1], arrs[2], n, dh.data_ptr(), _rowmajor(dh, "dh"),
                                        wt.data_ptr(), _rowmajor(wt, "wt"), _ptr(gfm), _ptr(sum_in), _ptr(arena),
                                        grad_arena.data_ptr(), int(accumulate), _ptr(dw),
                                        _rowmajor(dw, "dw") if dw is not None else 0, ws.data_ptr(), nbytes.value, _stream()),
               "rp_embed_grad_tiny")


_TINY_ARRAYS: dict = {}


def embed_grad_gemm(sorted_keys, sorted_pos, B: int, D: int, dh, wt, dx, gfm, sum_in, arena, grad_arena, accumulate: bool,
                    skip_fields: int = 0):
    """rp_embed_grad_gemm: the segmented reduce with the consuming Linear's dgrad formed inside (dh [B,64], wt = W1^T).
    skip_fields: bit f set = the pairs of field f are left out (field-major positions only; embed_grad_tiny's tables)."""
    _req(grad_arena, torch.float32, "grad_arena")
    _req(dh, torch.float32, "dh")
    _req(wt, torch.float32, "wt")
    ldx = _rowmajor(dx, "dx") if dx is not None else 0
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_reduce_workspace_bytes(sorted_keys.numel(), D, C.byref(nbytes)),
           "rp_embed_grad_reduce_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    with _Timed("embed_grad_gemm", f"D={D}"):
        _check(lib().rp_embed_grad_gemm(sorted_keys.data_ptr(), sorted_pos.data_ptr(), sorted_keys.numel(), B, D,
                                        dh.data_ptr(), _rowmajor(dh, "dh"), wt.data_ptr(), _rowmajor(wt, "wt"), _ptr(dx), ldx,
                                        _ptr(gfm), _ptr(sum_in), _ptr(arena), grad_arena.data_ptr(), int(accumulate),
                                        skip_fields, ws.data_ptr(), nbytes.value, _stream()), "rp_embed_grad_gemm")


def embed_grad_seg_fits(D: int, hidden: int, dh) -> bool:
    return bool(lib().rp_embed_grad_seg_fits(D, hidden, _rowmajor(dh, "dh"))) and dh.data_ptr() % 16 == 0


_SEG_ROWS: dict = {}


def embed_grad_seg(sorted_keys, sorted_pos, B: int, D: int, dh, w, gfm, sum_in, arena, grad_arena, accumulate: bool,
                   skip_fields: int = 0, field_rows=None, dw=None, keep=None):
    """rp_embed_grad_seg: the first layer's whole backward on the embedding columns, segment-sum first — the table
    gradient rows (dgrad of `w` [64, K] formed per run of equal keys + the FM term) AND, with dw [64, K], the embedding
    columns of the layer's weight gradient from the table rows the launch reads anyway (no stored activation).
    field_rows: table sizes per field (scheduling only).  Field-major positions (the single-device path)."""
    _req(grad_arena, torch.float32, "grad_arena")
    _req(dh, torch.float32, "dh")
    _req(w, torch.float32, "w")
    n = sorted_keys.numel()
    fr = None
    if field_rows is not None:
        ck = tuple(field_rows)
        fr = _SEG_ROWS.get(ck)
        if fr is None:
            fr = _SEG_ROWS[ck] = (C.c_int64 * len(ck))(*ck)
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_seg_workspace_bytes(n, B, D, C.byref(nbytes)), "rp_embed_grad_seg_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    if keep is not None:
        keep.append(ws)
    with _Timed("embed_grad_seg", f"D={D}"):  # (algorithmic bytes / flops: bench.py knows the unique-row count)
        _check(lib().rp_embed_grad_seg(sorted_keys.data_ptr(), sorted_pos.data_ptr(), n, B, D, dh.data_ptr(), _rowmajor(dh, "dh"),
                                       w.data_ptr(), _rowmajor(w, "w"), _ptr(gfm), _ptr(sum_in), arena.data_ptr(),
                                       grad_arena.data_ptr(), int(accumulate), skip_fields, fr, _ptr(dw),
                                       _rowmajor(dw, "dw") if dw is not None else 0, ws.data_ptr(), nbytes.value, _stream()),
               "rp_embed_grad_seg")


_SMP_FIELDS: dict = {}


def embed_grad_smp_fits(D: int, hidden: int, dh) -> bool:
    return bool(lib().rp_embed_grad_smp_fits(D, hidden, _rowmajor(dh, "dh"))) and dh.data_ptr() % 16 == 0


def _smp_fields(fields):
    """`fields` = [(field, first arena row, rows), ...] (or plain field indices for the mark launch) -> ctypes arrays"""
    ck = tuple((int(f[0]), int(f[1]), int(f[2])) if isinstance(f, (tuple, list)) else (int(f), 0, 0) for f in fields)
    arr = _SMP_FIELDS.get(ck)
    if arr is None:
        n = len(ck)
        arr = _SMP_FIELDS[ck] = ((C.c_int32 * n)(*[t[0] for t in ck]), (C.c_int64 * n)(*[t[1] for t in ck]),
                                 (C.c_int64 * n)(*[t[2] for t in ck]))
    return arr


def embed_grad_smp_mark(sorted_keys, sorted_pos, B: int, fields, out=None):
    """rp_embed_grad_smp_mark: (dupq, dupkeys, scratch) — which pairs of the big tables `fields` share their table row with
    another pair of the batch, numbered compactly in sorted order, from the sorted pair list (depends on the batch's ids only)."""
    _req(sorted_keys, torch.int32, "sorted_keys")
    _req(sorted_pos, torch.int32, "sorted_pos")
    nf = len(fields)
    if out is None:
        nsc = _sz(0)
        _check(lib().rp_embed_grad_smp_mark_scratch(B, nf, C.byref(nsc)), "rp_embed_grad_smp_mark_scratch")
        out = (torch.empty((nf * B,), dtype=torch.int32, device=sorted_keys.device),
               torch.empty((nf * B,), dtype=torch.int32, device=sorted_keys.device),
               torch.empty((nsc.value,), dtype=torch.int32, device=sorted_keys.device))
    with _Timed("embed_grad_smp_mark", f"{nf} fields", 16 * nf * B):
        _check(lib().rp_embed_grad_smp_mark(sorted_keys.data_ptr(), sorted_pos.data_ptr(), sorted_keys.numel(), B,
                                            _smp_fields(fields)[0], nf, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                            _stream()), "rp_embed_grad_smp_mark")
    return out


def embed_grad_smp(keys, marks, B: int, F: int, fields, dh, w, gfm, sum_in, arena, grad_arena, accumulate: bool, dw=None,
                   keep=None, phases: int = 3, ws=None):
    """rp_embed_grad_smp: the first layer's backward on the embedding columns of the big tables `fields`, sample-major (the
    table rows' gradient incl. the FM term + those fields' columns of dw [64, K]); `fields` = [(field, first arena row,
    rows), ...] ascending; `marks` = embed_grad_smp_mark's pair.  phases = 1: the main launch only, 2: the launches behind
    it (same `ws`, returned by the phase-1 call), 3: both.  -> the workspace."""
    _req(keys, torch.int32, "keys")
    _req(dh, torch.float32, "dh")
    _req(w, torch.float32, "w")
    _req(grad_arena, torch.float32, "grad_arena")
    nf = len(fields)
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_smp_workspace_bytes(B, nf, C.byref(nbytes)), "rp_embed_grad_smp_workspace_bytes")
    if ws is None:
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
        if keep is not None:
            keep.append(ws)
    fa = _smp_fields(fields)
    with _Timed("embed_grad_smp" if phases & 1 else "embed_grad_smp_behind", f"{nf} fields"):
        _check(lib().rp_embed_grad_smp(keys.data_ptr(), marks[0].data_ptr(), marks[1].data_ptr(), B, F, fa[0], fa[1], fa[2], nf,
                                       dh.data_ptr(), _rowmajor(dh, "dh"), w.data_ptr(), _rowmajor(w, "w"), _ptr(gfm),
                                       _ptr(sum_in), arena.data_ptr(), grad_arena.data_ptr(), int(accumulate), _ptr(dw),
                                       _rowmajor(dw, "dw") if dw is not None else 0, phases, ws.data_ptr(), nbytes.value, _stream()),
               "rp_embed_grad_smp")
    return ws


def embed_grad_reduce_rows(keys, rows, grad_arena, accumulate: bool):
    """rp_embed_grad_reduce_rows: grad_arena[key] (+)= the sum of rows[i] over a key-sorted list (key -1: no entry)"""
    _req(keys, torch.int32, "keys")
    _req(rows, torch.float32, "rows")
    n, D = keys.numel(), rows.shape[1]
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_reduce_workspace_bytes(n, D, C.byref(nbytes)), "rp_embed_grad_reduce_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    _check(lib().rp_embed_grad_reduce_rows(keys.data_ptr(), rows.data_ptr(), n, D, grad_arena.data_ptr(), int(accumulate),
                                           ws.data_ptr(), nbytes.value, _stream()), "rp_embed_grad_reduce_rows")


def embed_grad_ss_mark(sorted_keys, B: int, skip_fields: int, out=None):
    """rp_embed_grad_ss_mark: (ustart, ukey, offs) — the unique-row lists of the fields rp_embed_grad_ss keeps (every field
    whose bit in skip_fields is clear), from the sorted keys alone"""
    _req(sorted_keys, torch.int32, "sorted_keys")
    n = sorted_keys.numel()
    F = n // B
    kept = sum(1 for f in range(F) if not (skip_fields >> f) & 1)
    if out is None:
        nr, no = _sz(0), _sz(0)
        _check(lib().rp_embed_grad_ss_mark_sizes(B, kept, C.byref(nr), C.byref(no)), "rp_embed_grad_ss_mark_sizes")
        out = (torch.empty((nr.value,), dtype=torch.int32, device=sorted_keys.device),
               torch.empty((nr.value,), dtype=torch.int32, device=sorted_keys.device),
               torch.empty((no.value,), dtype=torch.int32, device=sorted_keys.device))
    with _Timed("embed_grad_ss_mark", f"{kept} fields", 12 * kept * B):
        _check(lib().rp_embed_grad_ss_mark(sorted_keys.data_ptr(), n, B, skip_fields, out[0].data_ptr(), out[1].data_ptr(),
                                           out[2].data_ptr(), _stream()), "rp_embed_grad_ss_mark")
    return out


def embed_grad_ss(sorted_keys, sorted_pos, B: int, D: int, dh, w, gfm, sum_in, arena, grad_arena, accumulate: bool,
                  skip_fields: int = 0, field_rows=None, dw=None, keep=None, phases: int = 3, ws=None, marks=None):
    """rp_embed_grad_ss: embed_grad_seg's work as a streaming segment-sum launch + a matrix launch over the unique rows (the
    mid-size tables' share of the first layer's backward); same arguments and results up to fp32 summation order.
    marks: embed_grad_ss_mark's triple for this sort and skip_fields (None: made inside the call).
    phases = 1: the segment-sum launch only, 2: the launches behind it (same `ws`), 3: both.  -> the workspace."""
    _req(grad_arena, torch.float32, "grad_arena")
    _req(dh, torch.float32, "dh")
    _req(w, torch.float32, "w")
    n = sorted_keys.numel()
    fr = None
    if field_rows is not None:
        ck = tuple(field_rows)
        fr = _SEG_ROWS.get(ck)
        if fr is None:
            fr = _SEG_ROWS[ck] = (C.c_int64 * len(ck))(*ck)
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_ss_workspace_bytes(n, B, D, skip_fields, C.byref(nbytes)), "rp_embed_grad_ss_workspace_bytes")
    if ws is None:
        ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
        if keep is not None:
            keep.append(ws)
    with _Timed("embed_grad_ss" if phases & 2 else "embed_grad_segsum", f"D={D}"):
        _check(lib().rp_embed_grad_ss(sorted_keys.data_ptr(), sorted_pos.data_ptr(), n, B, D, dh.data_ptr(), _rowmajor(dh, "dh"),
                                      w.data_ptr(), _rowmajor(w, "w"), _ptr(gfm), _ptr(sum_in), arena.data_ptr(),
                                      grad_arena.data_ptr(), int(accumulate), skip_fields, fr, _ptr(dw),
                                      _rowmajor(dw, "dw") if dw is not None else 0,
                                      None if marks is None else marks[0].data_ptr(), None if marks is None else marks[1].data_ptr(),
                                      None if marks is None else marks[2].data_ptr(), phases, ws.data_ptr(), nbytes.value, _stream()),
               "rp_embed_grad_ss")
    return ws


def zeros(shape, dtype, device):
    """torch.zeros through the library's own fill launch (rp_fill_words): inside a recorded step an ATen fill kernel would
    keep the step from replaying as a launch plan"""
    t = torch.empty(shape, dtype=dtype, device=device)
    nb = t.numel() * t.element_size()
    if nb % 4 != 0 or t.data_ptr() % 16 != 0:
        return t.zero_()
    if nb:
        _check(lib().rp_fill_words(t.data_ptr(), nb // 4, 0, _stream()), "rp_fill_words")
    return t


def zero_rows(keys, D: int, grad_arena):
    with _Timed("zero_rows"):
        _check(lib().rp_zero_rows(keys.data_ptr(), keys.numel(), D, grad_arena.data_ptr(), _stream()), "rp_zero_rows")


def linear_fwd(a, w, bias, act: int = ACT_NONE, aux=None, K: Optional[int] = None, out=None):
    """out[M,N] = act(a[:, :K] @ w[N,K]^T + bias). `a` may be wider than K (padded rows)."""
    _req(a, torch.float32, "a")
    _req(w, torch.float32, "w")
    lda, ldw = _rowmajor(a, "a"), _rowmajor(w, "w")
    M, N = a.shape[0], w.shape[0]
    K = w.shape[1] if K is None else K
    if a.shape[1] < K:
        raise RuntimeError(f"linear_fwd: a has {a.shape[1]} columns, need {K}")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ldo = _rowmajor(out, "out")
    ldaux = _rowmajor(aux, "aux") if aux is not None else 0
    with _Timed("linear_fwd", f"{M}x{N}x{K}", 4 * (M * K + N * K + M * N * (2 if aux is not None else 1)), 2 * M * N * K):
        _check(lib().rp_linear_fwd(a.data_ptr(), lda, w.data_ptr(), ldw, _ptr(bias), out.data_ptr(), ldo, M, N, K, act,
                               _ptr(aux), ldaux, _stream()), "rp_linear_fwd")
    return out


def linear_fwd_rowadd(a, w, row_scale, row_add, add_cols: int, out) -> bool:
    """out = a @ w^T + row_scale[:, None] * tile(row_add)[:, :add_cols]; False when the fused kernel does not cover
    the shape (nothing was launched)."""
    M, K = a.shape
    N = w.shape[0]
    if not (get_matmul_precision() != "fp32" and K <= 64 and K % 4 == 0 and M % 128 == 0 and M >= 128 and N % 64 == 0
            and add_cols % 64 == 0 and 0 < add_cols <= N and row_add.shape[1] == 64 and a.stride(0) % 4 == 0
            and w.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0 and out.shape[1] >= N):
        return False
    with _Timed("linear_fwd", f"{M}x{N}x{K}+rowadd", 4 * (M * K + N * K + M * N + M * 65), 2 * M * N * K):
        _check(lib().rp_linear_fwd_rowadd(a.data_ptr(), _rowmajor(a, "a"), w.data_ptr(), _rowmajor(w, "w"), out.data_ptr(),
                                          _rowmajor(out, "out"), M, N, K, row_scale.data_ptr(), row_add.data_ptr(),
                                          _rowmajor(row_add, "row_add"), add_cols, _stream()), "rp_linear_fwd_rowadd")
    return True


def linear_wgrad(dy, x, K: int, dw=None, db=None, accumulate: bool = False, want_bias: bool = True, keep=None):
    """dw[N,K] = dy[M,N]^T @ x[:, :K]; db[N] = colsum(dy).  keep: a list that receives the launch's workspace (a caller that
    lets the launch run beside later ones keeps it alive until they are joined)."""
    _req(dy, torch.float32, "dy")
    _req(x, torch.float32, "x")
    M, N = dy.shape
    lddy, ldx = _rowmajor(dy, "dy"), _rowmajor(x, "x")
    if dw is None:
        dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    if db is None and want_bias:
        db = torch.empty((N,), dtype=torch.float32, device=dy.device)
    nbytes = _sz(0)
    _check(lib().rp_linear_wgrad_workspace_bytes(M, N, K, C.byref(nbytes)), "rp_linear_wgrad_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dy.device)
    if keep is not None:
        keep.append(ws)
    with _Timed("linear_wgrad", f"{M}x{N}x{K}", 4 * (M * N + M * K + N * K), 2 * M * N * K):
        _check(lib().rp_linear_wgrad(dy.data_ptr(), lddy, x.data_ptr(), ldx, dw.data_ptr(), _rowmajor(dw, "dw"), _ptr(db),
                                 M, N, K, int(accumulate), ws.data_ptr(), nbytes.value, _stream()), "rp_linear_wgrad")
    return dw, db


def linear_wgrad_xbf16(dy, x16, K: int, want_bias: bool = True, keep=None):
    """rp_linear_wgrad_xbf16: dw[N, K] = dy^T @ x16[:, :K] with the activation stored as bfloat16 (bf16-storage training)"""
    _req(dy, torch.float32, "dy")
    _req(x16, torch.bfloat16, "x16")
    M, N = dy.shape
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_linear_wgrad_workspace_bytes(M, N, K, C.byref(nbytes)), "rp_linear_wgrad_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dy.device)
    if keep is not None:
        keep.append(ws)
    with _Timed("linear_wgrad_xbf16", f"{M}x{N}x{K}", 4 * (M * N + N * K) + 2 * M * K, 2 * M * N * K):
        _check(lib().rp_linear_wgrad_xbf16(dy.data_ptr(), _rowmajor(dy, "dy"), x16.data_ptr(), _rowmajor(x16, "x16"),
                                           dw.data_ptr(), K, _ptr(db), M, N, K, 0, ws.data_ptr(), nbytes.value, _stream()),
               "rp_linear_wgrad_xbf16")
    return dw, db


def linear_wgrad_gather_fits(M: int, N: int, K: int, Kg: int) -> bool:
    return bool(lib().rp_linear_wgrad_gather_fits(M, N, K, Kg))


def linear_wgrad_gather(dy, arena, keys, Kg: int, xd, K: int, want_bias: bool = True):
    """dw [N, K] = dy^T @ X with X[:, :Kg] gathered from the arena through `keys` ([F * M] int32, field-major) and
    X[:, Kg:K] = xd[:, :K - Kg] (rp_linear_wgrad_gather)"""
    _req(dy, torch.float32, "dy")
    _req(arena, torch.float32, "arena")
    _req(keys, torch.int32, "keys")
    M, N = dy.shape
    assert keys.numel() == (Kg // 64) * M and arena.shape[1] == 64
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    db = torch.empty((N,), dtype=torch.float32, device=dy.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_linear_wgrad_workspace_bytes(M, N, K, C.byref(nbytes)), "rp_linear_wgrad_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dy.device)
    with _Timed("linear_wgrad_gather", f"{M}x{N}x{K}", 4 * (M * N + M * K + N * K), 2 * M * N * K):
        _check(lib().rp_linear_wgrad_gather(dy.data_ptr(), _rowmajor(dy, "dy"), arena.data_ptr(), keys.data_ptr(), Kg, _ptr(xd),
                                            _rowmajor(xd, "xd") if xd is not None else 0, dw.data_ptr(), K, _ptr(db), M, N, K,
                                            0, ws.data_ptr(), nbytes.value, _stream()), "rp_linear_wgrad_gather")
    return dw, db


def transpose(w, rows_out: Optional[int] = None):
    """w [R, C] -> [C, R]; with rows_out > C the result has rows_out rows, the extra ones zero (a dgrad GEMM on it
    then writes exact zeros into the padding columns of a padded activation gradient)."""
    _req(w, torch.float32, "w")
    R, Cc = w.shape
    R4 = (R + 3) // 4 * 4  # 16-byte aligned rows (the GEMM then fetches them with dwordx4, unguarded)
    rows = max(Cc, rows_out or 0)
    out = torch.empty((rows, R4), dtype=torch.float32, device=w.device)[:, :R]
    with _Timed("transpose", f"{R}x{Cc}", 4 * (R * Cc + rows * R)):
        _check(lib().rp_transpose(w.data_ptr(), _rowmajor(w, "w"), out.data_ptr(), R4, R, Cc, rows, _stream()),
               "rp_transpose")
    return out


def transpose_copy(w, rows_out: int, ld_copy: int):
    """(transpose(w, rows_out), copy_rows(w, ld_copy)) from one launch (rp_transpose_copy)"""
    _req(w, torch.float32, "w")
    R, Cc = w.shape
    R4 = (R + 3) // 4 * 4
    rows = max(Cc, rows_out or 0)
    out = torch.empty((rows, R4), dtype=torch.float32, device=w.device)[:, :R]
    buf = torch.empty((R, ld_copy), dtype=torch.float32, device=w.device)
    with _Timed("transpose_copy", f"{R}x{Cc}", 4 * (2 * R * Cc + rows * R)):
        _check(lib().rp_transpose_copy(w.data_ptr(), _rowmajor(w, "w"), out.data_ptr(), R4, R, Cc, rows, buf.data_ptr(), ld_copy,
                                       _stream()), "rp_transpose_copy")
    return out, buf[:, :Cc]


def pieces_ld(K: int, np_: int) -> int:
    """bf16 elements per row of the interleaved-pieces layout (include/rec_pangu_hip.h: rp_pieces_ld)"""
    return int(lib().rp_pieces_ld(int(K), int(np_)))


def pieces_pack(x, np_: int = 2, K: Optional[int] = None, out=None):
    """fp32 [M, >=K] -> bf16 pieces [M, rp_pieces_ld(K, np)] in the interleaved layout (rp_pieces_pack)"""
    _req(x, torch.float32, "x")
    M = x.shape[0]
    K = x.shape[1] if K is None else K
    ldo = pieces_ld(K, np_)
    if out is None:
        out = torch.empty((M, ldo), dtype=torch.bfloat16, device=x.device)
    with _Timed("pieces_pack", f"{M}x{K}x{np_}", 4 * M * K + 2 * M * ldo):
        _check(lib().rp_pieces_pack(x.data_ptr(), _rowmajor(x, "x"), M, K, np_, out.data_ptr(), _rowmajor(out, "out"), _stream()),
               "rp_pieces_pack")
    return out


def linear_fwd_pieces(a_pc, w_pc, bias, K: int, np_: int = 2, act: int = ACT_NONE, aux=None, out=None):
    """out[M,N] = act(a . w^T + bias) on pre-split operands (pieces_pack layout); np = 2 is bit-identical to linear_fwd under
    RP_MATMUL_BF16X3, np = 1 is the plain-bf16 product"""
    _req(a_pc, torch.bfloat16, "a_pc")
    _req(w_pc, torch.bfloat16, "w_pc")
    M, N = a_pc.shape[0], w_pc.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a_pc.device)
    ldaux = _rowmajor(aux, "aux") if aux is not None else 0
    with _Timed("linear_fwd_pieces", f"{M}x{N}x{K}x{np_}", 2 * (M + N) * pieces_ld(K, np_) + 4 * M * N * (2 if aux is not None else 1),
                2 * M * N * K):
        _check(lib().rp_linear_fwd_pieces(a_pc.data_ptr(), _rowmajor(a_pc, "a_pc"), w_pc.data_ptr(), _rowmajor(w_pc, "w_pc"),
                                          _ptr(bias), out.data_ptr(), _rowmajor(out, "out"), M, N, K, np_, act, _ptr(aux), ldaux,
                                          _stream()), "rp_linear_fwd_pieces")
    return out


def copy_rows(w, ld_out: int):
    """a copy of the 2-D tensor w with row stride ld_out >= w.shape[1] (rp_copy_rows); returns the [R, C] view of it"""
    _req(w, torch.float32, "w")
    R, Cc = w.shape
    buf = torch.empty((R, ld_out), dtype=torch.float32, device=w.device)
    with _Timed("copy_rows", f"{R}x{Cc}", 8 * R * Cc):
        _check(lib().rp_copy_rows(w.data_ptr(), _rowmajor(w, "w"), buf.data_ptr(), ld_out, R, Cc, _stream()), "rp_copy_rows")
    return buf[:, :Cc]


def copy_rows_to(src, dst):
    """dst[r, :] = src[r, :] for 2-D fp32 tensors of one shape with unit inner stride, any row strides (rp_copy_rows): packs a
    matrix into a column block of a wider one, or a column block out into a contiguous matrix"""
    _req(src, torch.float32, "src")
    _req(dst, torch.float32, "dst")
    if src.dim() != 2 or src.shape != dst.shape or src.stride(1) != 1 or dst.stride(1) != 1:
        raise RuntimeError("copy_rows_to: two 2-D tensors of one shape with unit inner stride")
    R, Cc = src.shape
    with _Timed("copy_rows", f"{R}x{Cc}", 8 * R * Cc):
        _check(lib().rp_copy_rows(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), R, Cc, _stream()), "rp_copy_rows")
    return dst


def add_rows_to(src, dst):
    """dst[r, :] += src[r, :] for 2-D fp32 tensors of one shape with unit inner stride, any row strides (rp_add_rows)"""
    _req(src, torch.float32, "src")
    _req(dst, torch.float32, "dst")
    if src.dim() != 2 or src.shape != dst.shape or src.stride(1) != 1 or dst.stride(1) != 1:
        raise RuntimeError("add_rows_to: two 2-D tensors of one shape with unit inner stride")
    R, Cc = src.shape
    with _Timed("add_rows", f"{R}x{Cc}", 12 * R * Cc):
        _check(lib().rp_add_rows(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.stride(0), R, Cc, _stream()), "rp_add_rows")
    return dst


class Marker:
    """Completion marker on the current stream (rp_marker_*): an event without timing and without a default event's
    system-scope fence.  The host may wait for it; it must not read device memory on its strength."""

    def __init__(self):
        h = _vp()
        _check(lib().rp_marker_create(C.byref(h)), "rp_marker_create")
        self._h = h

    def record(self):
        _check(lib().rp_marker_record(self._h, _stream()), "rp_marker_record")

    def synchronize(self):
        _check(lib().rp_marker_wait(self._h), "rp_marker_wait")

    def __del__(self):
        try:
            if self._h is not None:
                lib().rp_marker_destroy(self._h)
        except Exception:
            pass


# ---- launch plans (csrc/plan.hip; used by graph_step.GraphedTrainStep) -------------------------------------------------
class LaunchPlan:
    """A recorded sequence of the library's kernel launches (rp_plan_*).  `with plan.recording(): ...` records every
    launch issued inside — from any thread — and `plan.replay()` re-issues them on the current stream."""

    def __init__(self):
        self._h = None
        self._streams = None
        self.nodes = self.side = self.streams = self.inline = 0
        self.host_calls = []  # what the caller issues itself between two segments of the replay (host marks); None = a cut
        self.seg_tags = [0]   # per segment: 0 = the replay's stream, 1 = the ahead stream (cut)
        self._tag = 0
        self.ahead_stream = None  # torch stream of the tag-1 segments (set by the owner of the plan)
        self._ev_fork = self._ev_join = None

    _recording = None  # the plan being recorded (host_call appends to it)

    def begin(self):
        h = _vp()
        _check(lib().rp_plan_begin(C.byref(h)), "rp_plan_begin")
        self._h = h
        LaunchPlan._recording = self

    @classmethod
    def host_call(cls, fn) -> bool:
        """Inside a recording: `fn` is work the library does not launch (a collective of the row-sharded path) — it is NOT
        issued now; the plan is cut here (rp_plan_host_mark) and every replay calls fn() between the two segments, on the
        stream of the segments around it.  -> True when the call was taken over (the caller skips issuing it), False outside
        a recording.  fn None: a plain cut (see cut)."""
        pl = cls._recording
        if pl is None or not cls.is_recording():
            return False
        k = _i32()
        _check(lib().rp_plan_host_mark(C.byref(k)), "rp_plan_host_mark")
        assert k.value == len(pl.host_calls)
        pl.host_calls.append(fn)
        pl.seg_tags.append(pl._tag)
        return True

    @classmethod
    def cut(cls, tag: int) -> bool:
        """Inside a recording of a plan that replays in segments (a row-sharded step): the segments recorded from here on —
        and the host calls between them — belong to stream `tag`: 0 = the replay's stream, 1 = the AHEAD stream (the next
        batch's route, id exchange and owner-side sort: work that depends on nothing the step computes; the replay forks it
        at its start and joins it at its end, LaunchPlan.replay)."""
        pl = cls._recording
        if pl is None or not cls.is_recording():
            return False
        pl._tag = int(tag)
        return cls.host_call(None)

    def end(self):
        LaunchPlan._recording = None
        if LaunchPlan._deferred and LaunchPlan.is_recording():
            LaunchPlan.join()  # (nobody joined the inline section after the last deferred launch was queued)
        LaunchPlan._deferred, LaunchPlan._kept, LaunchPlan._ahead_keep = [], [], []
        _check(lib().rp_plan_end(self._h), "rp_plan_end")
        a, b, c = _i32(), _i32(), _i32()
        _check(lib().rp_plan_info(self._h, C.byref(a), C.byref(b), C.byref(c)), "rp_plan_info")
        self.nodes, self.side, self.streams = a.value, b.value, c.value
        _check(lib().rp_plan_inline_count(self._h, C.byref(a)), "rp_plan_inline_count")
        self.inline = a.value

    def set_streams(self, side, side2):
        """torch streams the side / inline sections are re-issued on (kept alive by the caller)"""
        self._streams = (side, side2)
        _check(lib().rp_plan_set_streams(self._h, _vp(side.cuda_stream) if side is not None else None,
                                         _vp(side2.cuda_stream) if side2 is not None else None), "rp_plan_set_streams")

    @staticmethod
    def section(k: int):
        _check(lib().rp_plan_section(k), "rp_plan_section")

    @staticmethod
    def fork_here():
        _check(lib().rp_plan_fork_here(), "rp_plan_fork_here")

    @staticmethod
    def fork2_mark():
        """the inline section's launches recorded from here on depend on the main stream as it is NOW (rp_plan_fork2_mark)"""
        _check(lib().rp_plan_fork2_mark(), "rp_plan_fork2_mark")

    # launches a recording step wants on the inline section (2) but has no hurry with: issued behind the next launches
    # recorded there (run_deferred), at the latest in front of the section's join; `keep` = the tensors they touch (the
    # capture's allocator would hand their memory to the launches recorded in between)
    _deferred: list = []
    _kept: list = []

    @classmethod
    def defer_side(cls, fn, keep=()):
        cls._deferred.append(fn)
        cls._kept.append(keep)

    @classmethod
    def run_deferred(cls):
        """issue the deferred launches now, on the inline section (the caller may already be inside it)"""
        if not cls._deferred:
            return
        todo, cls._deferred = cls._deferred, []
        _check(lib().rp_plan_section(2), "rp_plan_section")
        try:
            for fn in todo:
                fn()
        finally:
            _check(lib().rp_plan_section(0), "rp_plan_section")

    @classmethod
    def settle(cls):
        """after the backward of a recording step: launches still waiting for the inline section are issued and joined NOW
        (their results are the optimizer's inputs; a step without the fused first-layer node never joined them)"""
        if cls._deferred and cls.is_recording():
            cls.join()

    @classmethod
    def join(cls):
        cls.run_deferred()
        _check(lib().rp_plan_join(), "rp_plan_join")
        cls._kept = []

    @classmethod
    def join_only(cls):
        """join the inline section as it stands: deferred launches stay queued, what they keep alive stays alive"""
        _check(lib().rp_plan_join(), "rp_plan_join")

    @staticmethod
    def side2_sync():
        """the inline section waits here for what the main stream holds at this point (rp_plan_side2_sync)"""
        _check(lib().rp_plan_side2_sync(), "rp_plan_side2_sync")

    @staticmethod
    def join_side():
        """the main stream waits here for the side section (rp_plan_join_side)"""
        _check(lib().rp_plan_join_side(), "rp_plan_join_side")

    # "catch-up ahead" (graph_step.GraphedTrainStep, recording only): the step catches up the NEXT batch's rows at its end,
    # beside the dense optimizer step — the first layer's backward then joins only what that launch must not overtake (the
    # tiny tables' gradient, which reads their rows) and leaves the rest of its side work for the optimizer's side section
    ahead = False
    _ahead_keep: list = []

    @staticmethod
    def is_recording() -> bool:
        return bool(lib().rp_plan_is_recording())

    def replay(self):
        if self.host_calls:  # segments, the caller's own work (collectives) in between
            s = _stream()
            tags, calls, nseg = self.seg_tags, self.host_calls, len(self.host_calls) + 1
            side = self.ahead_stream if any(tags) else None
            if any(tags) and side is None:
                tags = [0] * nseg  # (no ahead stream given: everything in recorded order on the replay's stream)
            if side is not None:
                # fork: the ahead segments depend on what was enqueued BEFORE this replay only (the next batch's staged ids,
                # the previous replay's reads of the buffers they rewrite)
                main = torch.cuda.current_stream()
                if self._ev_fork is None:
                    self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
                self._ev_fork.record(main)
                side.wait_event(self._ev_fork)
                s1 = _vp(side.cuda_stream)
                # the ahead segments in front of their first host call (the route: launches only) are issued first — beside
                # the step from its start; their collectives follow the step's own (RCCL runs collectives in issue order: an
                # id exchange waiting for the route must not sit in front of the row / gradient exchanges of the step)
                first_side = next(k for k in range(nseg) if tags[k] == 1)
                rc = lib().rp_plan_replay_segment(self._h, first_side, s1)
                if rc != 0:
                    _check(rc, "rp_plan_replay_segment")
            else:
                first_side = -1
            for k in range(nseg):  # the replay's own stream: segments and the host calls between two of them
                if tags[k] == 0:
                    rc = lib().rp_plan_replay_segment(self._h, k, s)
                    if rc != 0:
                        _check(rc, "rp_plan_replay_segment")
                    if k < nseg - 1 and tags[k + 1] == 0 and calls[k] is not None:
                        calls[k]()
            if side is not None:
                with torch.cuda.stream(side):
                    for k in range(first_side, nseg):
                        if tags[k] != 1:
                            continue
                        if k > first_side:
                            if calls[k - 1] is not None and tags[k - 1] == 1:
                                calls[k - 1]()
                            rc = lib().rp_plan_replay_segment(self._h, k, s1)
                            if rc != 0:
                                _check(rc, "rp_plan_replay_segment")
                self._ev_join.record(side)
                main.wait_event(self._ev_join)
            return
        rc = lib().rp_plan_replay(self._h, _stream())
        if rc != 0:
            _check(rc, "rp_plan_replay")

    # ---- live timing of one launch inside replayed steps (bench.py: a replay runs no python between its launches) ------
    def set_probe(self, launch: int):
        """bracket launch `launch` (0 .. nodes - 1, recorded order; -1 = off) of the following replays with a timing-event pair"""
        _check(lib().rp_plan_set_probe(self._h, launch), "rp_plan_set_probe")

    def probe_ms(self) -> float:
        ms = C.c_float(0)
        _check(lib().rp_plan_probe_ms(self._h, C.byref(ms)), "rp_plan_probe_ms")
        return ms.value

    def bind_inputs(self, addrs) -> int:
        """remember where the recorded launch arguments hold the addresses `addrs` (the static input buffers the step was
        recorded on); -> number of argument words found (rp_plan_bind_inputs)"""
        arr = (C.c_uint64 * len(addrs))(*addrs)
        n = _i32()
        _check(lib().rp_plan_bind_inputs(self._h, arr, len(addrs), C.byref(n)), "rp_plan_bind_inputs")
        self._n_inputs = len(addrs)
        self._in_arr = (C.c_uint64 * len(addrs))()
        return n.value

    def bind_report(self, tensors):
        """(sites per input, interior words): how often the recorded launch arguments hold exactly tensors[i].data_ptr(), and
        how many hold an address INSIDE one of the tensors — derived pointers that set_inputs cannot re-point
        (rp_plan_bind_report)"""
        n = len(tensors)
        addrs = (C.c_uint64 * n)(*[t.data_ptr() for t in tensors])
        sizes = (C.c_uint64 * n)(*[t.numel() * t.element_size() for t in tensors])
        sites = (C.c_int32 * n)()
        inner = _i32()
        _check(lib().rp_plan_bind_report(self._h, addrs, sizes, n, sites, C.byref(inner)), "rp_plan_bind_report")
        return list(sites), inner.value

    def set_inputs(self, tensors):
        """the next replays read input i from tensors[i] (rp_plan_set_inputs)"""
        a = self._in_arr
        for i, t in enumerate(tensors):
            a[i] = t.data_ptr()
        _check(lib().rp_plan_set_inputs(self._h, a, self._n_inputs), "rp_plan_set_inputs")

    def slowest_call(self, reset: bool = True):
        """(kind, node, ms) of the slowest HIP call the replays issued since the last reset (kind: launch / record / wait)"""
        k, n, ms = C.c_int(-1), C.c_int(-1), C.c_double(0)
        _check(lib().rp_plan_slowest_call(self._h, C.byref(k), C.byref(n), C.byref(ms), 1 if reset else 0), "rp_plan_slowest_call")
        return (("launch", "record", "wait")[k.value] if k.value >= 0 else None), n.value, ms.value

    def launch_name(self, launch: int):
        """(kernel name as rocprofv3 prints it, section: 0 main stream, 1 side (next batch's sort), 2 inline side)"""
        buf, sec = C.create_string_buffer(512), _i32()
        _check(lib().rp_plan_launch_name(self._h, launch, buf, 512, C.byref(sec)), "rp_plan_launch_name")
        return buf.value.decode("utf-8", "replace"), sec.value

    def destroy(self):
        if self._h is not None and _lib is not None:
            _lib.rp_plan_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


_COPY_PLANS: dict = {}


# Tensors the wrappers below allocate while a `holding()` block is open stay referenced by its list.  Why: inside a stream
# capture the allocator hands a freed block to the NEXT allocation of the same capture — fine for launches that replay in
# recorded order on one stream, wrong for a group of launches the replay runs on another stream BESIDE the rest (the
# row-sharded step's look-ahead: graph_step).  Their temporaries — also the ones a wrapper allocates and drops internally —
# must outlive the recording of everything they run beside.
_HOLDING = None


class holding:
    def __enter__(self):
        global _HOLDING
        self.prev, self.items = _HOLDING, []
        _HOLDING = self.items
        return self.items

    def __exit__(self, *exc):
        global _HOLDING
        _HOLDING = self.prev
        return False


def _held(*tensors):
    if _HOLDING is not None:
        _HOLDING.extend(t for t in tensors if t is not None)


def make_side_stream(device, role: str = "sort") -> "torch.cuda.Stream":
    """a stream for work that runs BESIDE the main stream (the next batch's sort, the side work of the first layer's backward):
    of the LOWEST priority the device offers (rp_stream_create_low, wrapped as an ExternalStream), so that the main stream's
    launches are dispatched first — measured in alternating runs on one box: 0.912 / 0.912 -> 0.903 / 0.887 ms per step in the
    20-step window, 0.879 -> 0.866 over 600 steps.  RP_SIDE_PRIORITY=normal: a plain torch stream (rounds 2-4)"""
    prio = os.environ.get("RP_SIDE_PRIORITY", "low")
    if role == "inline":  # (the first layer's side work is waited for by the optimizer; the sort by nothing in its step)
        prio = os.environ.get("RP_SIDE2_PRIORITY", prio)
    if prio == "low":
        with torch.cuda.device(device):
            h = _vp()
            _check(lib().rp_stream_create_low(C.byref(h)), "rp_stream_create_low")
        return torch.cuda.ExternalStream(h.value, device=device)
    if prio == "high":
        return torch.cuda.Stream(device=device, priority=-1)
    return torch.cuda.Stream(device=device)


def multi_copy(dst: Sequence[torch.Tensor], src: Sequence[torch.Tensor]) -> bool:
    """dst[i].copy_(src[i]) for every i in ONE launch (rp_multi_copy).  Only for same-shape, same-dtype, contiguous
    tensors on the current device — returns False (nothing done) otherwise, the caller then copies tensor by tensor."""
    n = len(dst)
    for d, s_ in zip(dst, src):
        if not (d.is_cuda and s_.is_cuda and d.device == s_.device and d.dtype == s_.dtype and d.shape == s_.shape
                and d.is_contiguous() and s_.is_contiguous()):
            return False
    # (the byte counts are part of the key: a re-allocation at the same addresses with another shape or dtype must not
    #  find the old sizes — ADVICE r4)
    key = (tuple(t.data_ptr() for t in dst) + tuple(t.data_ptr() for t in src)
           + tuple(t.numel() * t.element_size() for t in dst))
    arrs = _COPY_PLANS.get(key)
    if arrs is None:
        if len(_COPY_PLANS) >= 256:
            _COPY_PLANS.clear()
        arrs = ((C.c_void_p * n)(*[t.data_ptr() for t in dst]), (C.c_void_p * n)(*[t.data_ptr() for t in src]),
                (C.c_uint64 * n)(*[t.numel() * t.element_size() for t in dst]))
        _COPY_PLANS[key] = arrs
    _check(lib().rp_multi_copy(arrs[0], arrs[1], arrs[2], n, _stream()), "rp_multi_copy")
    return True


class CopyList:
    """rp_multi_copy into a FIXED set of destination tensors (a captured step's static input buffers): the destination
    pointers and byte counts are converted once, a call only fills the source pointers (the general multi_copy validates and
    keys ~40 tensor pairs per call: a third of the host time of a replayed step)."""

    def __init__(self, dst: Sequence[torch.Tensor]):
        self.n = len(dst)
        self.shapes = [(t.shape, t.dtype) for t in dst]
        self.dst = (C.c_void_p * self.n)(*[t.data_ptr() for t in dst])
        self.bytes = (C.c_uint64 * self.n)(*[t.numel() * t.element_size() for t in dst])
        self.src = (C.c_void_p * self.n)()
        self.ok = all(t.is_cuda and t.is_contiguous() for t in dst)

    def __call__(self, src: Sequence[torch.Tensor]) -> bool:
        """copy src[i] into the i-th destination; False (nothing done) unless every source is a contiguous device tensor of the
        destination's shape and dtype"""
        if not self.ok or len(src) != self.n:
            return False
        a = self.src
        for i, t in enumerate(src):
            if not (t.is_cuda and t.is_contiguous() and (t.shape, t.dtype) == self.shapes[i]):
                return False
            a[i] = t.data_ptr()
        _check(lib().rp_multi_copy(self.dst, a, self.bytes, self.n, _stream()), "rp_multi_copy")
        return True


def graph_node_counts(raw_graph: int):
    """(kernel nodes, other nodes) of a captured hipGraph (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph())"""
    k, o = _i32(), _i32()
    _check(lib().rp_graph_node_counts(_vp(raw_graph), C.byref(k), C.byref(o)), "rp_graph_node_counts")
    return k.value, o.value


def relu_bwd(dy, act_out):
    M, N = dy.shape
    out = torch.empty((M, N), dtype=torch.float32, device=dy.device)
    with _Timed("relu_bwd", f"{M}x{N}", 12 * M * N):
        _check(lib().rp_relu_bwd(dy.data_ptr(), _rowmajor(dy, "dy"), act_out.data_ptr(), _rowmajor(act_out, "act_out"),
                             out.data_ptr(), N, M, N, _stream()), "rp_relu_bwd")
    return out


def act_bwd(dy, act_out, act: int):
    """dy * act'(.) through the activation's output (rp_act_bwd; act = ACT_RELU / TANH / SIGMOID / LEAKY)"""
    M, N = dy.shape
    out = torch.empty((M, N), dtype=torch.float32, device=dy.device)
    with _Timed("act_bwd", f"{M}x{N}", 12 * M * N):
        _check(lib().rp_act_bwd(dy.data_ptr(), _rowmajor(dy, "dy"), act_out.data_ptr(), _rowmajor(act_out, "act_out"),
                                out.data_ptr(), N, M, N, act, _stream()), "rp_act_bwd")
    return out


def act_fwd(x, act: int):
    """act(x) as a launch of its own (rp_act_fwd), 2-D x"""
    M, N = x.shape
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _Timed("act_fwd", f"{M}x{N}", 8 * M * N):
        _check(lib().rp_act_fwd(x.data_ptr(), _rowmajor(x, "x"), out.data_ptr(), N, M, N, act, _stream()), "rp_act_fwd")
    return out


def crossnet_fwd(x0, d: int, W, Bv, wfc=None, bfc=None, want_x: bool = True):
    """X_L (and/or logit = X_L . wfc + bfc) of the L-layer CrossNet; returns (xout, logit, s)."""
    _req(x0, torch.float32, "x0")
    B, L = x0.shape[0], W.shape[0]
    dev = x0.device
    xout = torch.empty((B, d), dtype=torch.float32, device=dev) if want_x else None
    logit = torch.empty((B, 1), dtype=torch.float32, device=dev) if wfc is not None else None
    s = torch.empty((B, L), dtype=torch.float32, device=dev)
    with _Timed("crossnet_fwd"):
        _check(lib().rp_crossnet_fwd(x0.data_ptr(), _rowmajor(x0, "x0"), d, L, W.data_ptr(), Bv.data_ptr(), _ptr(wfc),
                                     _ptr(bfc), _ptr(xout), d, _ptr(logit), s.data_ptr(), B, _stream()),
               "rp_crossnet_fwd")
    return xout, logit, s


def crossnet_bwd_rows(x0, d: int, W, wfc, s, g_x, g_logit):
    """-> dx0 [B, x0.shape[1]] (columns >= d zeroed), V [B, 2L+2] (see rp_crossnet_bwd_rows)."""
    B, L = x0.shape[0], W.shape[0]
    dx0 = torch.empty_like(x0)
    ldg = _rowmajor(g_x, "g_x") if g_x is not None else 0
    # the vectorised kernel writes whole dwordx4 up to the leading dimension (exact zeros beyond d); only the unaligned form
    # leaves the padding columns to a fill (an ATen launch: it kept a DCN step from replaying as a launch plan)
    vec = (x0.stride(0) % 4 == 0 and dx0.stride(0) % 4 == 0 and x0.data_ptr() % 16 == 0 and dx0.data_ptr() % 16 == 0
           and (g_x is None or (ldg % 4 == 0 and g_x.data_ptr() % 16 == 0)) and x0.shape[1] <= (d + 255) // 256 * 256)
    if x0.shape[1] > d and not vec:
        dx0[:, d:].zero_()
    V = torch.empty((B, 2 * L + 2), dtype=torch.float32, device=x0.device)
    with _Timed("crossnet_bwd_rows"):
        _check(lib().rp_crossnet_bwd_rows(x0.data_ptr(), _rowmajor(x0, "x0"), d, L, W.data_ptr(), _ptr(wfc),
                                          s.data_ptr(), _ptr(g_x), ldg, _ptr(g_logit), dx0.data_ptr(),
                                          _rowmajor(dx0, "dx0"), V.data_ptr(), B, _stream()), "rp_crossnet_bwd_rows")
    return dx0, V


def crossnet_param_grads(P, cs, W, Bv, wfc, colg):
    """(dW [L, d], dB [L, d], dwfc [d] or None) of the CrossNet from the skinny weight gradient P = V^T X_0 and the column
    sums cs of V (rp_crossnet_param_grads)"""
    L, d = W.shape
    dW = torch.empty((L, d), dtype=torch.float32, device=W.device)
    dB = torch.empty((L, d), dtype=torch.float32, device=W.device)
    dwfc = torch.empty((d,), dtype=torch.float32, device=W.device) if wfc is not None else None
    with _Timed("crossnet_param_grads", f"{L}x{d}"):
        _check(lib().rp_crossnet_param_grads(P.data_ptr(), _rowmajor(P, "P"), cs.data_ptr(), W.data_ptr(), Bv.data_ptr(), _ptr(wfc),
                                             _ptr(colg), L, d, dW.data_ptr(), dB.data_ptr(), _ptr(dwfc), _stream()),
               "rp_crossnet_param_grads")
    return dW, dB, dwfc


def cin_layer_fwd(x0, xp, W, bias, H: int, M: int, D: int, want_out: bool, want_pool: bool):
    """One CIN layer.  x0 [B, >=H*D], xp [B, >=M*D] (pass the same tensor for the first layer), W [O, H*M].
    -> (out [B,O,D] or None, pooled [B,O] or None)."""
    _req(x0, torch.float32, "x0")
    _req(xp, torch.float32, "xp")
    B, O = x0.shape[0], W.shape[0]
    out = torch.empty((B, O, D), dtype=torch.float32, device=x0.device) if want_out else None
    pooled = torch.empty((B, O), dtype=torch.float32, device=x0.device) if want_pool else None
    with _Timed("cin_layer_fwd"):
        _check(lib().rp_cin_layer_fwd(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"),
                                      W.data_ptr(), _ptr(bias), _ptr(out), _ptr(pooled), O, H, M, O, D, B, _stream()),
               "rp_cin_layer_fwd")
    return out, pooled


def cin_layer_bwd(x0, xp, W, H: int, M: int, D: int, g_out, g_pool, want_bias: bool):
    """Gradients of one CIN layer: -> (dx0 like x0, dxp [B, M*D] or None when xp is x0, dW like W, dbias [O] or None)."""
    B, O = x0.shape[0], W.shape[0]
    same = xp is x0
    dx0 = torch.empty_like(x0)
    if x0.shape[1] > H * D:
        dx0[:, H * D:].zero_()
    dxp = None if same else torch.empty((B, M * D), dtype=torch.float32, device=x0.device)
    ldgp = _rowmajor(g_pool, "g_pool") if g_pool is not None else 0
    with _Timed("cin_layer_bwd_x"):
        _check(lib().rp_cin_layer_bwd_x(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"),
                                        W.data_ptr(), _ptr(g_out), _ptr(g_pool), ldgp, dx0.data_ptr(),
                                        _rowmajor(dx0, "dx0"), 0, _ptr(dxp), M * D, H, M, O, D, B, _stream()),
               "rp_cin_layer_bwd_x")
    dW = torch.empty_like(W)
    db = torch.empty((O,), dtype=torch.float32, device=x0.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_cin_layer_bwd_w_workspace_bytes(B, H, M, O, C.byref(nbytes)), "rp_cin_layer_bwd_w_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x0.device)
    with _Timed("cin_layer_bwd_w"):
        _check(lib().rp_cin_layer_bwd_w(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"),
                                        _ptr(g_out), _ptr(g_pool), ldgp, dW.data_ptr(), _ptr(db), H, M, O, D, B,
                                        ws.data_ptr(), nbytes.value, _stream()), "rp_cin_layer_bwd_w")
    return dx0, dxp, dW, db


def field_attention_fits(T: int, Din: int, H: int, a: int, has_res: bool) -> bool:
    return bool(lib().rp_field_attention_fits(T, Din, H, a, int(has_res)))


def field_attention_fwd(x, W, T: int, Din: int, H: int, a: int, has_res: bool, scale: float):
    """x [B, >=T*Din] (tokens contiguous per sample) -> out [B, T, H*a]."""
    _req(x, torch.float32, "x")
    _req(W, torch.float32, "W")
    B = x.shape[0]
    out = torch.empty((B, T, H * a), dtype=torch.float32, device=x.device)
    with _Timed("field_attention_fwd"):
        _check(lib().rp_field_attention_fwd(x.data_ptr(), _rowmajor(x, "x"), W.data_ptr(), T, Din, H, a, int(has_res),
                                            scale, out.data_ptr(), B, _stream()), "rp_field_attention_fwd")
    return out


def field_attention_bwd(x, W, T: int, Din: int, H: int, a: int, has_res: bool, scale: float, gout, want_dx: bool):
    B = x.shape[0]
    dx = None
    if want_dx:
        dx = torch.empty_like(x)
        if x.shape[1] > T * Din:
            dx[:, T * Din:].zero_()
    dW = torch.empty_like(W)
    nbytes = _sz(0)
    _check(lib().rp_field_attention_bwd_workspace_bytes(B, T, Din, H, a, int(has_res), C.byref(nbytes)),
           "rp_field_attention_bwd_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x.device)
    with _Timed("field_attention_bwd"):
        _check(lib().rp_field_attention_bwd(x.data_ptr(), _rowmajor(x, "x"), W.data_ptr(), T, Din, H, a, int(has_res),
                                            scale, gout.data_ptr(), _ptr(dx), _rowmajor(dx, "dx") if want_dx else 0,
                                            dW.data_ptr(), B, ws.data_ptr(), nbytes.value, _stream()),
               "rp_field_attention_bwd")
    return dx, dW


def cin_bs_fits(H: int, M: int, D: int) -> bool:
    return bool(lib().rp_cin_bs_fits(H, M, D))


def bf16_pieces(w3: torch.Tensor) -> torch.Tensor:
    """[O, R, C] fp32 (R, C <= 32) -> [O, 3, 32, 32] bf16: (hi, mid, lo) with w = hi + mid + lo (+ 2^-24 |w|), zero
    padded — the operand format of the split-bf16 matrix-core kernels (round-to-nearest-even conversions, exactly
    what v_cvt_pk_bf16_f32 does in the kernels that split on the fly)."""
    O, R, Cc = w3.shape
    pad = torch.zeros((O, 32, 32), dtype=torch.float32, device=w3.device)
    pad[:, :R, :Cc] = w3
    hi = pad.to(torch.bfloat16)
    r1 = pad - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo], dim=1).contiguous()


def cin_bs_fwd(x0, xp, wp, bias, H: int, M: int, O: int, D: int, want_out: bool, want_pool: bool):
    B = x0.shape[0]
    out = torch.empty((B, O, D), dtype=torch.float32, device=x0.device) if want_out else None
    pooled = torch.empty((B, O), dtype=torch.float32, device=x0.device) if want_pool else None
    with _Timed("cin_bs_fwd"):
        _check(lib().rp_cin_bs_fwd(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"), wp.data_ptr(),
                                   _ptr(bias), H, M, O, D, _ptr(out), _ptr(pooled), B, _stream()), "rp_cin_bs_fwd")
    return out, pooled


def accumulate(dst, src):
    """dst += src (contiguous fp32 tensors of one size; rp_accumulate)"""
    _req(dst, torch.float32, "dst")
    _req(src, torch.float32, "src")
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    with _Timed("accumulate", None, 12 * dst.numel()):
        _check(lib().rp_accumulate(dst.data_ptr(), src.data_ptr(), dst.numel(), _stream()), "rp_accumulate")
    return dst


def cin_bs_bwd_x(xk, wp, g_out, g_pool, R: int, Cn: int, O: int, D: int, like, out=None):
    """-> dx shaped like `like` ([B, >= R*D], zero beyond R*D), or written into `out` ([B, R*D] view, any leading
    dimension)."""
    B = xk.shape[0]
    dx = out if out is not None else torch.empty_like(like)
    if out is None and like.shape[1] > R * D:
        dx[:, R * D:].zero_()
    with _Timed("cin_bs_bwd_x"):
        _check(lib().rp_cin_bs_bwd_x(xk.data_ptr(), _rowmajor(xk, "xk"), wp.data_ptr(), _ptr(g_out), _ptr(g_pool), R, Cn, O,
                                     D, dx.data_ptr(), _rowmajor(dx, "dx"), B, _stream()), "rp_cin_bs_bwd_x")
    return dx


def cin_bs_bwd_w(x0, xp, g_out, g_pool, H: int, M: int, O: int, D: int, want_bias: bool):
    """dW [O, H*M], dbias [O] or None on the bf16 matrix core (rp_cin_bs_bwd_w); g_pool must be packed [B, O]."""
    B = x0.shape[0]
    dW = torch.empty((O, H * M), dtype=torch.float32, device=x0.device)
    db = torch.empty((O,), dtype=torch.float32, device=x0.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_cin_bs_bwd_w_workspace_bytes(B, O, C.byref(nbytes)), "rp_cin_bs_bwd_w_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x0.device)
    with _Timed("cin_bs_bwd_w"):
        _check(lib().rp_cin_bs_bwd_w(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"), _ptr(g_out),
                                     _ptr(g_pool), H, M, O, D, dW.data_ptr(), _ptr(db), B, ws.data_ptr(), nbytes.value,
                                     _stream()), "rp_cin_bs_bwd_w")
    return dW, db


def cin_pair_fits(H: int, O: int, D: int) -> bool:
    return bool(lib().rp_cin_pair_fits(H, O, D))


def _bf16_split3(full):
    hi = full.to(torch.bfloat16)
    r1 = full - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack((hi, mid, lo)).contiguous()


def _cin_pair_ws(W3):
    O, H, _ = W3.shape
    iu = torch.triu_indices(H, H, device=W3.device)  # row-major upper triangle
    return W3[:, iu[0], iu[1]] + W3[:, iu[1], iu[0]] * (iu[0] != iu[1]).to(W3.dtype)


def cin_pair_pieces_torch(W3, transposed: bool = False):
    """the torch formulation of cin_pair_pieces (rounds 3-5; kept as the tests' reference for rp_cin_pair_pieces)"""
    O = W3.shape[0]
    ws = _cin_pair_ws(W3)
    npair = ws.shape[1]
    if transposed:
        full = torch.zeros(((npair + 127) // 128 * 128, 128), dtype=torch.float32, device=W3.device)
        full[:npair, :O] = ws.t()
    else:
        full = torch.zeros((128, (npair + 31) // 32 * 32), dtype=torch.float32, device=W3.device)
        full[:O, :npair] = ws
    return _bf16_split3(full)


def cin_pair_pieces(W3, transposed: bool = False, both: bool = False):
    """W [O, H, H] -> the symmetric pair weights Ws[o, (h<=m)] = W[o,h,m] + W[o,m,h] (W[o,h,h] on the diagonal) as bf16
    pieces: [3, 128, KP] (rp_cin_pair_fwd's wsp, KP = pairs rounded up to 32) or, transposed, [3, KPT, 128]
    (rp_cin_pair_bwd_x's wst, KPT = pairs rounded up to 128); both: (wsp, wst) from ONE launch (rp_cin_pair_pieces, round 6:
    the fold + split were ~18 torch launches per layout and step)."""
    _req(W3, torch.float32, "W")
    O, H = W3.shape[0], W3.shape[1]
    W3 = W3.contiguous()
    npair = H * (H + 1) // 2
    dev = W3.device
    wsp = torch.empty((3, 128, (npair + 31) // 32 * 32), dtype=torch.bfloat16, device=dev) if (both or not transposed) else None
    wst = torch.empty((3, (npair + 127) // 128 * 128, 128), dtype=torch.bfloat16, device=dev) if (both or transposed) else None
    _check(lib().rp_cin_pair_pieces(W3.data_ptr(), O, H, _ptr(wsp), _ptr(wst), _stream()), "rp_cin_pair_pieces")
    return (wsp, wst) if both else (wst if transposed else wsp)


def cin_head_params_fwd(WL, bL, c, H: int, M: int):
    """-> (vt [M, 32], vb [1]): V^T = (c . W_L)^T zero padded and c . b_L of the collapsed last CIN layer (rp_cin_head_params_fwd)"""
    O = WL.shape[0]
    vt = torch.empty((M, 32), dtype=torch.float32, device=WL.device)
    vb = torch.empty((1,), dtype=torch.float32, device=WL.device)
    _check(lib().rp_cin_head_params_fwd(WL.data_ptr(), _ptr(bL), c.data_ptr(), O, H, M, vt.data_ptr(), vb.data_ptr(), _stream()),
           "rp_cin_head_params_fwd")
    return vt, vb


def cin_head_params_bwd(WL, bL, c, dV, sg, D: int, H: int, M: int):
    """-> (dWL [O, H M], dbL [O] or None, dc [O]) of the collapsed last layer's weights from dV [H, M] and sg = sum g
    (rp_cin_head_params_bwd)"""
    O = WL.shape[0]
    dWL = torch.empty((O, H * M), dtype=torch.float32, device=WL.device)
    dbL = torch.empty((O,), dtype=torch.float32, device=WL.device) if bL is not None else None
    dc = torch.empty((O,), dtype=torch.float32, device=WL.device)
    _check(lib().rp_cin_head_params_bwd(WL.data_ptr(), _ptr(bL), c.data_ptr(), dV.data_ptr(), sg.data_ptr(), float(D), O, H, M,
                                        dWL.data_ptr(), _ptr(dbL), dc.data_ptr(), _stream()), "rp_cin_head_params_bwd")
    return dWL, dbL, dc


def add_scalars(out, a, scale: float, b0=None):
    """out[i] += scale * a[0] + (b0[0] if b0 is given), in place (rp_add_scalars)"""
    assert out.is_contiguous() and out.dtype == torch.float32
    _check(lib().rp_add_scalars(out.data_ptr(), out.numel(), a.data_ptr(), float(scale), _ptr(b0), _stream()), "rp_add_scalars")
    return out


def sum_all(x):
    """-> [1] = the sum of x's elements in a fixed order (rp_sum_all)"""
    assert x.is_contiguous() and x.dtype == torch.float32
    out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _check(lib().rp_sum_all(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "rp_sum_all")
    return out


_PAIR_LISTS = {}


def cin_pair_lists(H: int, device):
    """(lstart, lent) of rp_cin_pair_bwd_x for H fields, built once per (H, device)."""
    key = (H, str(device))
    if key not in _PAIR_LISTS:
        pairs = [(h, m) for h in range(H) for m in range(h, H)]  # row-major upper triangle
        ntile = (len(pairs) + 127) // 128
        lstart, lent = [0], []
        for tile in range(ntile):
            for half in range(2):
                lo = tile * 128 + half * 64
                loc = {}
                for pl, (h, m) in enumerate(pairs[lo:lo + 64]):
                    loc.setdefault(h, []).append(pl | (m << 8))
                    loc.setdefault(m, []).append(pl | (h << 8))  # (h == m: the diagonal pair twice)
                for h in range(H):
                    lent.extend(loc.get(h, []))
                    lstart.append(len(lent))
        _PAIR_LISTS[key] = (torch.tensor(lstart, dtype=torch.int32, device=device),
                            torch.tensor(lent if lent else [0], dtype=torch.int32, device=device))
    return _PAIR_LISTS[key]


def cin_pair_bwd_x(x0, wst, g_out, g_pool, H: int, O: int, D: int, like, into=None):
    """-> dX_0 shaped like `like` ([B, >= H*D], zero beyond H*D) in the pair form (rp_cin_pair_bwd_x).  into: a [B, >= H*D]
    tensor the gradient is ADDED to and that is returned (the collapsed last layer's gradient of X_0: no pass of its own)"""
    B = x0.shape[0]
    if into is not None:
        _req(into, torch.float32, "into")
        assert into.shape[0] == B and into.shape[1] >= H * D and into.stride(1) == 1
        dx = into
    else:
        dx = torch.empty_like(like)
        if like.shape[1] > H * D:
            dx[:, H * D:].zero_()
    lstart, lent = cin_pair_lists(H, x0.device)
    with _Timed("cin_pair_bwd_x"):
        _check(lib().rp_cin_pair_bwd_x(x0.data_ptr(), _rowmajor(x0, "x0"), wst.data_ptr(), _ptr(g_out), _ptr(g_pool),
                                       lstart.data_ptr(), lent.data_ptr(), H, O, D, dx.data_ptr(), _rowmajor(dx, "dx"), B,
                                       1 if into is not None else 0, _stream()), "rp_cin_pair_bwd_x")
    return dx


def cin_pair_fwd(x0, wsp, bias, H: int, O: int, D: int, want_out: bool, want_pool: bool):
    B = x0.shape[0]
    out = torch.empty((B, O, D), dtype=torch.float32, device=x0.device) if want_out else None
    pooled = torch.empty((B, O), dtype=torch.float32, device=x0.device) if want_pool else None
    with _Timed("cin_pair_fwd"):
        _check(lib().rp_cin_pair_fwd(x0.data_ptr(), _rowmajor(x0, "x0"), wsp.data_ptr(), _ptr(bias), H, O, D, _ptr(out),
                                     _ptr(pooled), B, _stream()), "rp_cin_pair_fwd")
    return out, pooled


def cin_pair_bwd_w(x0, g_out, g_pool, H: int, O: int, D: int, want_bias: bool):
    """first-layer dW [O, H*H], dbias [O] or None in the symmetric pair form (rp_cin_pair_bwd_w); O <= 128."""
    B = x0.shape[0]
    dW = torch.empty((O, H * H), dtype=torch.float32, device=x0.device)
    db = torch.empty((O,), dtype=torch.float32, device=x0.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_cin_pair_bwd_w_workspace_bytes(B, H, O, C.byref(nbytes)), "rp_cin_pair_bwd_w_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x0.device)
    with _Timed("cin_pair_bwd_w"):
        _check(lib().rp_cin_pair_bwd_w(x0.data_ptr(), _rowmajor(x0, "x0"), _ptr(g_out), _ptr(g_pool), H, O, D, dW.data_ptr(),
                                       _ptr(db), B, ws.data_ptr(), nbytes.value, _stream()), "rp_cin_pair_bwd_w")
    return dW, db


def cin_layer_bwd_w(x0, xp, W, H: int, M: int, D: int, g_out, g_pool, want_bias: bool):
    """dW like W, dbias [O] or None (rp_cin_layer_bwd_w alone)."""
    B, O = x0.shape[0], W.shape[0]
    ldgp = _rowmajor(g_pool, "g_pool") if g_pool is not None else 0
    dW = torch.empty_like(W)
    db = torch.empty((O,), dtype=torch.float32, device=x0.device) if want_bias else None
    nbytes = _sz(0)
    _check(lib().rp_cin_layer_bwd_w_workspace_bytes(B, H, M, O, C.byref(nbytes)), "rp_cin_layer_bwd_w_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x0.device)
    with _Timed("cin_layer_bwd_w"):
        _check(lib().rp_cin_layer_bwd_w(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"),
                                        _ptr(g_out), _ptr(g_pool), ldgp, dW.data_ptr(), _ptr(db), H, M, O, D, B,
                                        ws.data_ptr(), nbytes.value, _stream()), "rp_cin_layer_bwd_w")
    return dW, db


def cin_last_fits(H: int, M: int, D: int) -> bool:
    return bool(lib().rp_cin_last_fits(H, M, D))


def cin_last_fwd(x0, xp, vt, H: int, M: int, D: int):
    """x0 [B, >=H*D], xp [B, >=M*D], vt [M, 32] (V^T zero padded) -> pooled [B, 1]."""
    _req(x0, torch.float32, "x0")
    B = x0.shape[0]
    pooled = torch.empty((B, 1), dtype=torch.float32, device=x0.device)
    with _Timed("cin_last_fwd"):
        _check(lib().rp_cin_last_fwd(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"), vt.data_ptr(),
                                     H, M, D, pooled.data_ptr(), B, _stream()), "rp_cin_last_fwd")
    return pooled


def cin_last_bwd(x0, xp, vt, g, H: int, M: int, D: int):
    """-> dx0 (shape of x0, zero beyond H*D), dxp (shape of xp), dV [H, M]."""
    B = x0.shape[0]
    dx0, dxp = torch.empty_like(x0), torch.empty_like(xp)
    if x0.shape[1] > H * D:
        dx0[:, H * D:].zero_()
    if xp.shape[1] > M * D:
        dxp[:, M * D:].zero_()
    with _Timed("cin_last_bwd_x"):
        _check(lib().rp_cin_last_bwd_x(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"),
                                       vt.data_ptr(), g.data_ptr(), H, M, D, dx0.data_ptr(), _rowmajor(dx0, "dx0"),
                                       dxp.data_ptr(), _rowmajor(dxp, "dxp"), B, _stream()), "rp_cin_last_bwd_x")
    dV = torch.empty((H, M), dtype=torch.float32, device=x0.device)
    nbytes = _sz(0)
    _check(lib().rp_cin_last_bwd_v_workspace_bytes(B, H, M, C.byref(nbytes)), "rp_cin_last_bwd_v_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=x0.device)
    with _Timed("cin_last_bwd_v"):
        _check(lib().rp_cin_last_bwd_v(x0.data_ptr(), _rowmajor(x0, "x0"), xp.data_ptr(), _rowmajor(xp, "xp"), g.data_ptr(),
                                       H, M, D, dV.data_ptr(), B, ws.data_ptr(), nbytes.value, _stream()),
               "rp_cin_last_bwd_v")
    return dx0, dxp, dV


def attention_core_fits(T: int, H: int, a: int) -> bool:
    return bool(lib().rp_attention_core_fits(T, H, a))


def attention_core_fwd(qkvr, nproj: int, xres, T: int, H: int, a: int, scale: float):
    """qkvr [B*T, nproj*H*a] (Q|K|V|R), xres [B*T, H*a] when nproj == 3 -> out [B*T, H*a], stats [B, H*T, 2]."""
    _req(qkvr, torch.float32, "qkvr")
    BT = qkvr.shape[0]
    B = BT // T
    out = torch.empty((BT, H * a), dtype=torch.float32, device=qkvr.device)
    stats = torch.empty((B, H * T, 2), dtype=torch.float32, device=qkvr.device)
    with _Timed("attention_core_fwd"):
        _check(lib().rp_attention_core_fwd(qkvr.data_ptr(), _rowmajor(qkvr, "qkvr"), nproj, _ptr(xres),
                                           _rowmajor(xres, "xres") if xres is not None else 0, T, H, a, scale,
                                           out.data_ptr(), stats.data_ptr(), B, _stream()), "rp_attention_core_fwd")
    return out, stats


def attention_core_bwd(qkvr, nproj: int, out, dout, stats, T: int, H: int, a: int, scale: float):
    BT = qkvr.shape[0]
    dqkvr = torch.empty_like(qkvr)
    dxres = torch.empty((BT, H * a), dtype=torch.float32, device=qkvr.device) if nproj == 3 else None
    with _Timed("attention_core_bwd"):
        _check(lib().rp_attention_core_bwd(qkvr.data_ptr(), _rowmajor(qkvr, "qkvr"), nproj, out.data_ptr(), dout.data_ptr(),
                                           stats.data_ptr(), T, H, a, scale, dqkvr.data_ptr(), _rowmajor(dqkvr, "dqkvr"),
                                           _ptr(dxres), H * a if dxres is not None else 0, BT // T, _stream()),
               "rp_attention_core_bwd")
    return dqkvr, dxres


def mmoe_combine_fwd(z, K: int, E: int, T: int):
    """z [B, >=K*E+T*E] -> (out [T,B,K], gate [B,T*E])."""
    _req(z, torch.float32, "z")
    B = z.shape[0]
    out = torch.empty((T, B, K), dtype=torch.float32, device=z.device)
    gate = torch.empty((B, T * E), dtype=torch.float32, device=z.device)
    with _Timed("mmoe_combine_fwd"):
        _check(lib().rp_mmoe_combine_fwd(z.data_ptr(), _rowmajor(z, "z"), K, E, T, out.data_ptr(), gate.data_ptr(), B,
                                         _stream()), "rp_mmoe_combine_fwd")
    return out, gate


def mmoe_combine_bwd(z, K: int, E: int, T: int, gate, dout):
    """dout [T,B,K] -> dz [B, K*E+T*E]."""
    B = z.shape[0]
    dz = torch.empty((B, K * E + T * E), dtype=torch.float32, device=z.device)
    with _Timed("mmoe_combine_bwd"):
        _check(lib().rp_mmoe_combine_bwd(z.data_ptr(), _rowmajor(z, "z"), K, E, T, gate.data_ptr(), dout.data_ptr(),
                                         dz.data_ptr(), K * E + T * E, B, _stream()), "rp_mmoe_combine_bwd")
    return dz


MATMUL_MODES = {"fp32": 0, "bf16": 1, "auto": 2, "bf16x3": 3, "bf16x6": 6}


def set_matmul_precision(mode: str):
    """GEMM matrix-core mode: 'auto' (default: per launch 'bf16x3' when matrix-core bound, 'bf16x6' when HBM-bound),
    'bf16x6' (fp32-faithful split-bf16), 'bf16x3', 'bf16', 'fp32' (f32 MFMA)."""
    _check(lib().rp_set_matmul_precision(MATMUL_MODES[mode]), "rp_set_matmul_precision")


def get_matmul_precision() -> str:
    v = lib().rp_get_matmul_precision()
    return {n: k for k, n in MATMUL_MODES.items()}[v]


def fm_pool_fwd(x2d, F: int, D: int, want_sum: bool, want_bi: bool):
    """x2d [B, >=F*D] -> (sum [B,1] or None, bi [B,D] or None)."""
    _req(x2d, torch.float32, "x")
    B = x2d.shape[0]
    out_sum = torch.empty((B, 1), dtype=torch.float32, device=x2d.device) if want_sum else None
    out_bi = torch.empty((B, D), dtype=torch.float32, device=x2d.device) if want_bi else None
    with _Timed("fm_pool_fwd"):
        _check(lib().rp_fm_pool_fwd(x2d.data_ptr(), _rowmajor(x2d, "x"), F, D, _ptr(out_sum), _ptr(out_bi), B,
                                    _stream()), "rp_fm_pool_fwd")
    return out_sum, out_bi


def fm_pool_bwd(x2d, F: int, D: int, g_sum, g_bi):
    dx = torch.empty_like(x2d)
    if x2d.shape[1] > F * D:
        dx[:, F * D:].zero_()
    with _Timed("fm_pool_bwd"):
        _check(lib().rp_fm_pool_bwd(x2d.data_ptr(), _rowmajor(x2d, "x"), F, D, _ptr(g_sum), _ptr(g_bi), dx.data_ptr(),
                                    _rowmajor(dx, "dx"), x2d.shape[0], _stream()), "rp_fm_pool_bwd")
    return dx


def _bn_ws(M, N, dev):
    nbytes = _sz(0)
    _check(lib().rp_batchnorm_workspace_bytes(M, N, C.byref(nbytes)), "rp_batchnorm_workspace_bytes")
    return torch.empty((nbytes.value,), dtype=torch.uint8, device=dev), nbytes.value


def batchnorm_train_fwd(x, gamma, beta, eps: float):
    """-> y, mean [N], var [N] (biased), rstd [N]"""
    _req(x, torch.float32, "x")
    M, N = x.shape
    dev = x.device
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    mean, var, rstd = (torch.empty((N,), dtype=torch.float32, device=dev) for _ in range(3))
    ws, nb = _bn_ws(M, N, dev)
    with _Timed("batchnorm_train_fwd", f"{M}x{N}", 16 * M * N):  # 3 reads + 1 write
        _check(lib().rp_batchnorm_train_fwd(x.data_ptr(), _rowmajor(x, "x"), _ptr(gamma), _ptr(beta), eps, y.data_ptr(),
                                            N, mean.data_ptr(), var.data_ptr(), rstd.data_ptr(), M, N, ws.data_ptr(),
                                            nb, _stream()), "rp_batchnorm_train_fwd")
    return y, mean, var, rstd


def batchnorm_train_bwd(x, dy, mean, rstd, gamma):
    M, N = x.shape
    dev = x.device
    dx = torch.empty((M, N), dtype=torch.float32, device=dev)
    dgamma, dbeta = (torch.empty((N,), dtype=torch.float32, device=dev) for _ in range(2))
    ws, nb = _bn_ws(M, N, dev)
    with _Timed("batchnorm_train_bwd", f"{M}x{N}", 20 * M * N):  # x, dy twice + dx
        _check(lib().rp_batchnorm_train_bwd(x.data_ptr(), _rowmajor(x, "x"), dy.data_ptr(), _rowmajor(dy, "dy"),
                                            mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), dx.data_ptr(), N,
                                            dgamma.data_ptr(), dbeta.data_ptr(), M, N, ws.data_ptr(), nb, _stream()),
               "rp_batchnorm_train_bwd")
    return dx, dgamma, dbeta


def batchnorm_apply(x, mean, rstd, gamma, beta):
    M, N = x.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _Timed("batchnorm_apply", f"{M}x{N}", 8 * M * N):
        _check(lib().rp_batchnorm_apply(x.data_ptr(), _rowmajor(x, "x"), mean.data_ptr(), rstd.data_ptr(), _ptr(gamma),
                                        _ptr(beta), y.data_ptr(), N, M, N, _stream()), "rp_batchnorm_apply")
    return y


def dice_gate_fwd(x, xhat, alpha):
    """y = x * (alpha + sigmoid(xhat) * (1 - alpha))  (rp_dice_gate_fwd)"""
    _req(x, torch.float32, "x")
    _req(xhat, torch.float32, "xhat")
    _req(alpha, torch.float32, "alpha")
    M, N = x.shape
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _Timed("dice_gate_fwd", f"{M}x{N}", 12 * M * N):
        _check(lib().rp_dice_gate_fwd(x.data_ptr(), _rowmajor(x, "x"), xhat.data_ptr(), _rowmajor(xhat, "xhat"),
                                      alpha.data_ptr(), y.data_ptr(), N, M, N, _stream()), "rp_dice_gate_fwd")
    return y


def dice_gate_bwd(x, xhat, alpha, dy):
    """-> (dx_direct, dxhat, dal) packed [M, N] each (rp_dice_gate_bwd); dalpha = column sums of dal"""
    M, N = x.shape
    dxd, dxh, dal = (torch.empty((M, N), dtype=torch.float32, device=x.device) for _ in range(3))
    with _Timed("dice_gate_bwd", f"{M}x{N}", 24 * M * N):
        _check(lib().rp_dice_gate_bwd(x.data_ptr(), _rowmajor(x, "x"), xhat.data_ptr(), _rowmajor(xhat, "xhat"),
                                      alpha.data_ptr(), dy.data_ptr(), _rowmajor(dy, "dy"), dxd.data_ptr(), dxh.data_ptr(),
                                      dal.data_ptr(), M, N, _stream()), "rp_dice_gate_bwd")
    return dxd, dxh, dal


def batchnorm_apply_bwd(dy, rstd, gamma):
    M, N = dy.shape
    dx = torch.empty((M, N), dtype=torch.float32, device=dy.device)
    with _Timed("batchnorm_apply_bwd", f"{M}x{N}", 8 * M * N):
        _check(lib().rp_batchnorm_apply_bwd(dy.data_ptr(), _rowmajor(dy, "dy"), rstd.data_ptr(), _ptr(gamma),
                                            dx.data_ptr(), N, M, N, _stream()), "rp_batchnorm_apply_bwd")
    return dx


def sigmoid_bce_fwd(addends: Sequence[torch.Tensor], label: Optional[torch.Tensor], apply_sigmoid: bool = True,
                    p_eps: float = 0.0, weight: float = 1.0, add_to: Optional[torch.Tensor] = None):
    B = addends[0].numel()
    for z in addends:
        _req(z, torch.float32, "logit")
        if z.numel() != B or not z.is_contiguous():
            raise RuntimeError("logit addends must be contiguous with B elements")
    dev = addends[0].device
    pred = torch.empty((B, 1), dtype=torch.float32, device=dev)
    loss = partial = None
    if label is not None:
        _req(label, torch.float32, "label")
        if label.numel() != B or not label.is_contiguous():
            raise RuntimeError("label must be contiguous float32 with B elements")
        partial = torch.empty((lib().rp_loss_partials(B),), dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev) if add_to is None else add_to
    elif add_to is not None:
        raise RuntimeError("sigmoid_bce_fwd: add_to needs a label")
    with _Timed("sigmoid_bce_fwd"):
        # add_to: the loss scalar of the tasks before this one (a multi-task loss): this launch ADDS its weighted mean
        fn = lib().rp_sigmoid_bce_fwd if add_to is None else lib().rp_sigmoid_bce_fwd_accum
        _check(fn(_ptr_array(addends), len(addends), int(apply_sigmoid), _ptr(label), B, p_eps, weight, pred.data_ptr(),
                  _ptr(partial), _ptr(loss), _stream()), "rp_sigmoid_bce_fwd")
    return pred, loss


def sigmoid_bce_bwd(pred, label, gloss, apply_sigmoid: bool = True, p_eps: float = 0.0, weight: float = 1.0):
    B = pred.numel()
    dz = torch.empty((B, 1), dtype=torch.float32, device=pred.device)
    gloss = gloss.reshape(1).contiguous()
    with _Timed("sigmoid_bce_bwd"):
        _check(lib().rp_sigmoid_bce_bwd(pred.data_ptr(), label.data_ptr(), gloss.data_ptr(), B, p_eps, weight,
                                    int(apply_sigmoid), dz.data_ptr(), _stream()), "rp_sigmoid_bce_bwd")
    return dz


_weight_epoch = 0


def weight_epoch() -> int:
    """bumped by every optimizer launch of this module: parameters change through raw pointers there, which torch's
    tensor version counters do not see (caches of derived weight layouts key on this)"""
    return _weight_epoch


def bump_weight_epoch() -> None:
    """a captured step was replayed: its optimizer kernels ran without any python"""
    global _weight_epoch
    _weight_epoch += 1


def counters_add(counters, delta: int = 1):
    """*c += delta for every device counter in `counters` (1..8 int32[1] tensors) in ONE launch (rp_counters_add)"""
    if len(counters) == 1:
        return counter_add(counters[0], delta)
    _check(lib().rp_counters_add(_ptr_array(counters), len(counters), delta, _stream()), "rp_counters_add")


def counter_add(counter, delta: int = 1):
    """*counter += delta on the current stream (device-resident step counters, see rp_counter_add)"""
    _check(lib().rp_counter_add(counter.data_ptr(), delta, _stream()), "rp_counter_add")


def adam_step(params, grads, ms, vs, lr, beta1, beta2, eps, step: int, zero_grad: bool, scalars=None, t_dev=None):
    """One fused launch per <=64 tensors; tensors must be contiguous fp32 on the same device.  t_dev (device int32[1],
    completed steps) + scalars (the per-step float2 table): the step number is read on the device (hipGraph replays)."""
    global _weight_epoch
    _weight_epoch += 1
    for i in range(0, len(params), MAX_FIELDS):
        ps, gs = params[i:i + MAX_FIELDS], grads[i:i + MAX_FIELDS]
        mm, vv = ms[i:i + MAX_FIELDS], vs[i:i + MAX_FIELDS]
        for t in (*ps, *gs, *mm, *vv):
            _req(t, torch.float32, "adam tensor")
            if not t.is_contiguous():
                raise RuntimeError("adam tensors must be contiguous")
        sizes = (C.c_int64 * len(ps))(*[p.numel() for p in ps])
        with _Timed("adam_step"):
            _check(lib().rp_adam_step(_ptr_array(ps), _ptr_array(gs), _ptr_array(mm), _ptr_array(vv), sizes, len(ps), lr,
                                  beta1, beta2, eps, step, int(zero_grad), _ptr(scalars), _ptr(t_dev), _stream()),
                   "rp_adam_step")


# ---- exact lazy dense Adam (arena rows) ---------------------------------------------------------------
def embed_keys(row_base, row_count, idx: List[torch.Tensor], err_flag, out=None):
    F, B = len(idx), idx[0].shape[0]
    keys = out if out is not None else torch.empty((F * B,), dtype=torch.int32, device=idx[0].device)
    assert keys.numel() == F * B and keys.dtype == torch.int32
    _held(keys)
    with _Timed("embed_keys"):
        _check(lib().rp_embed_keys(row_base.data_ptr(), row_count.data_ptr(), _ptr_array(idx), F, B, keys.data_ptr(),
                                   err_flag.data_ptr(), _stream()), "rp_embed_keys")
    return keys


def shard_keys(row_base, row_count, idx: List[torch.Tensor], world: int, lbits: int, err_flag):
    """composite (owner << lbits | local row) int32 keys of a batch's row requests, p = f*B + b (rp_shard_keys)."""
    F, B = len(idx), idx[0].shape[0]
    keys = torch.empty((F * B,), dtype=torch.int32, device=idx[0].device)
    _held(keys)
    with _Timed("shard_keys"):
        _check(lib().rp_shard_keys(row_base.data_ptr(), row_count.data_ptr(), _ptr_array(idx), F, B, world, lbits,
                                   keys.data_ptr(), err_flag.data_ptr(), _stream()), "rp_shard_keys")
    return keys


def route_build(sorted_keys, sorted_pos, world: int, lbits: int, out=None):
    """-> (slot_sorted int32 [n], slot_of_pair int64 [n], uniq_rows int64 [n] (first counts[world] valid),
    counts int64 [world+1]) from the sorted composite keys (rp_route_build).  out = (slot_sorted, slot_of_pair): persistent
    buffers to write into (the prepared lookup of a recorded step's static batch)"""
    n = sorted_keys.numel()
    dev = sorted_keys.device
    if out is not None:
        slot_sorted, slot_of_pair = out
        assert slot_sorted.shape == (n,) and slot_sorted.dtype == torch.int32 and slot_of_pair.shape == (n,) \
            and slot_of_pair.dtype == torch.int64
    else:
        slot_sorted = torch.empty((n,), dtype=torch.int32, device=dev)
        slot_of_pair = torch.empty((n,), dtype=torch.int64, device=dev)
    uniq_rows = torch.empty((n,), dtype=torch.int64, device=dev)
    counts = torch.empty((world + 1,), dtype=torch.int64, device=dev)
    nbytes = _sz(0)
    _check(lib().rp_route_workspace_bytes(n, world, C.byref(nbytes)), "rp_route_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
    _held(slot_sorted, slot_of_pair, uniq_rows, counts, ws)
    with _Timed("route_build"):
        _check(lib().rp_route_build(ws.data_ptr(), nbytes.value, sorted_keys.data_ptr(), sorted_pos.data_ptr(), n, world,
                                    lbits, slot_sorted.data_ptr(), slot_of_pair.data_ptr(), uniq_rows.data_ptr(),
                                    counts.data_ptr(), _stream()), "rp_route_build")
    return slot_sorted, slot_of_pair, uniq_rows, counts


def route_pad(sorted_keys, sorted_pos, world: int, lbits: int, capacity: int, counts, slot_sorted, slot_of_pair, err_flag,
              out=None):
    """fixed-capacity form of a route (rp_route_pad): slot_sorted / slot_of_pair are rewritten in place / overwritten;
    -> rows_padded int64 [world * capacity] (local rows to ask each owner for, unused slots 0); out: a persistent buffer for it"""
    if out is not None:
        assert out.shape == (world * capacity,) and out.dtype == torch.int64
        rows_padded = out
        _check(lib().rp_fill_words(rows_padded.data_ptr(), rows_padded.numel() * 2, 0, _stream()), "rp_fill_words")
    else:
        rows_padded = zeros((world * capacity,), torch.int64, sorted_keys.device)
    _held(rows_padded)
    with _Timed("route_pad"):
        _check(lib().rp_route_pad(sorted_keys.data_ptr(), sorted_pos.data_ptr(), sorted_keys.numel(), world, lbits, capacity,
                                  counts.data_ptr(), slot_sorted.data_ptr(), slot_of_pair.data_ptr(),
                                  rows_padded.data_ptr(), err_flag.data_ptr(), _stream()), "rp_route_pad")
    return rows_padded


def route_field_major(sorted_keys, sorted_pos, slot_sorted, B: int, world: int, lbits: int, out=None):
    """-> (slot_fm, pos_fm): the route's sorted (slot, position) list moved into FIELD-major order (rp_route_field_major) —
    what rp_embed_grad_seg takes as its sorted pair list when the rows came from more than one owner"""
    n = sorted_keys.numel()
    dev = sorted_keys.device
    if out is not None:
        slot_fm, pos_fm = out
        assert slot_fm.shape == (n,) and pos_fm.shape == (n,) and slot_fm.dtype == pos_fm.dtype == torch.int32
    else:
        slot_fm = torch.empty((n,), dtype=torch.int32, device=dev)
        pos_fm = torch.empty((n,), dtype=torch.int32, device=dev)
    delta = torch.empty((world * (n // B),), dtype=torch.int64, device=dev)
    _held(slot_fm, pos_fm, delta)
    with _Timed("route_field_major"):
        _check(lib().rp_route_field_major(sorted_keys.data_ptr(), sorted_pos.data_ptr(), slot_sorted.data_ptr(), n, B, world, lbits,
                                          slot_fm.data_ptr(), pos_fm.data_ptr(), delta.data_ptr(), _stream()),
               "rp_route_field_major")
    return slot_fm, pos_fm


def batchnorm_update_running(mean, var, bn, M: int):
    """bn.running_mean / running_var / num_batches_tracked after one training forward over M rows (rp_batchnorm_update_running)"""
    for t in (mean, var, bn.running_mean, bn.running_var):
        _req(t, torch.float32, "statistics")
    nbt = bn.num_batches_tracked
    if nbt is not None and (nbt.dtype is not torch.int64 or not nbt.is_cuda):
        raise RuntimeError("batchnorm_update_running: num_batches_tracked must be an int64 device tensor")
    _check(lib().rp_batchnorm_update_running(mean.data_ptr(), var.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                             _ptr(nbt), -1.0 if bn.momentum is None else float(bn.momentum), M, mean.numel(),
                                             _stream()), "rp_batchnorm_update_running")


def batchnorm_colsum(x, center=None):
    """column sums of x, or of (x - center)^2 (rp_batchnorm_colsum) -> [N]"""
    _req(x, torch.float32, "x")
    M, N = x.shape
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    ws, nb = _bn_ws(M, N, x.device)
    with _Timed("batchnorm_colsum", f"{M}x{N}", 4 * M * N):
        _check(lib().rp_batchnorm_colsum(x.data_ptr(), _rowmajor(x, "x"), _ptr(center), out.data_ptr(), M, N, ws.data_ptr(),
                                         nb, _stream()), "rp_batchnorm_colsum")
    return out


def batchnorm_bwd_sums(x, dy, mean, rstd):
    """-> (dgamma [N] = sum dy * xhat, dbeta [N] = sum dy) over the local rows (rp_batchnorm_bwd_sums)"""
    M, N = x.shape
    dgamma, dbeta = (torch.empty((N,), dtype=torch.float32, device=x.device) for _ in range(2))
    ws, nb = _bn_ws(M, N, x.device)
    with _Timed("batchnorm_bwd_sums", f"{M}x{N}", 8 * M * N):
        _check(lib().rp_batchnorm_bwd_sums(x.data_ptr(), _rowmajor(x, "x"), dy.data_ptr(), _rowmajor(dy, "dy"),
                                           mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), M, N,
                                           ws.data_ptr(), nb, _stream()), "rp_batchnorm_bwd_sums")
    return dgamma, dbeta


def batchnorm_bwd_apply(x, dy, mean, rstd, gamma, mean_dy, mean_dyx):
    M, N = x.shape
    dx = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _Timed("batchnorm_bwd_apply", f"{M}x{N}", 12 * M * N):
        _check(lib().rp_batchnorm_bwd_apply(x.data_ptr(), _rowmajor(x, "x"), dy.data_ptr(), _rowmajor(dy, "dy"),
                                            mean.data_ptr(), rstd.data_ptr(), _ptr(gamma), mean_dy.data_ptr(),
                                            mean_dyx.data_ptr(), dx.data_ptr(), N, M, N, _stream()),
               "rp_batchnorm_bwd_apply")
    return dx


def mlp_tail_fits(n_hidden: int, width: int, hin) -> bool:
    return bool(lib().rp_mlp_tail_fits(n_hidden, width, _rowmajor(hin, "hin"))) and hin.data_ptr() % 16 == 0


def _i64_array(vals):
    return (C.c_int64 * max(len(vals), 1))(*vals)


def _opt_ptr_array(tensors):
    arr = (C.c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else _ptr(t)
    return arr


def mlp_tail_fwd(hin, Ws, bs, w_out, b_out):
    """hin [M,64] -> (logit [M,1], hidden outputs [M,64] x len(Ws)) in one launch (rp_mlp_tail_fwd)."""
    _req(hin, torch.float32, "hin")
    M, L = hin.shape[0], len(Ws)
    hs = [torch.empty((M, 64), dtype=torch.float32, device=hin.device) for _ in range(L)]
    logit = torch.empty((M, 1), dtype=torch.float32, device=hin.device)
    with _Timed("mlp_tail_fwd", f"{M}x64x{L}", 4 * M * (64 * (L + 1) + 1), 2 * M * (64 * 64 * L + 64)):
        _check(lib().rp_mlp_tail_fwd(hin.data_ptr(), _rowmajor(hin, "hin"), L, _ptr_array(Ws), _i64_array([_rowmajor(w, "W") for w in Ws]),
                                     _opt_ptr_array(bs), _ptr_array(hs), w_out.data_ptr(), _ptr(b_out), logit.data_ptr(), M,
                                     _stream()), "rp_mlp_tail_fwd")
    return logit, hs


def mlp_tail_fwd_bce(hin, Ws, bs, w_out, b_out, addends, label, p_eps: float = 0.0, weight: float = 1.0):
    """the tail's forward with the loss head inside (rp_mlp_tail_fwd_bce): -> (pred [M,1], loss [] , hidden outputs).
    The scalar is the only thing the partial sums are for and nobody on the device reads it: inside a recorded launch plan
    rp_loss_finish is issued on the plan's inline side section (deferred behind the next launches recorded there)."""
    _req(hin, torch.float32, "hin")
    _req(label, torch.float32, "label")
    M, L = hin.shape[0], len(Ws)
    dev = hin.device
    hs = [torch.empty((M, 64), dtype=torch.float32, device=dev) for _ in range(L)]
    pred = torch.empty((M, 1), dtype=torch.float32, device=dev)
    n_part = lib().rp_mlp_tail_loss_partials(M)
    partial = torch.empty((n_part,), dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    for t in addends:
        _req(t, torch.float32, "addend")
        if t.numel() != M or not t.is_contiguous():
            raise RuntimeError("mlp_tail_fwd_bce: every addend is a contiguous [M] / [M, 1] logit")
    if label.numel() != M or not label.is_contiguous():
        raise RuntimeError("mlp_tail_fwd_bce: label must be a contiguous [M]")
    with _Timed("mlp_tail_fwd", f"{M}x64x{L}+bce", 4 * M * (64 * (L + 1) + 3), 2 * M * (64 * 64 * L + 64)):
        _check(lib().rp_mlp_tail_fwd_bce(hin.data_ptr(), _rowmajor(hin, "hin"), L, _ptr_array(Ws),
                                         _i64_array([_rowmajor(w, "W") for w in Ws]), _opt_ptr_array(bs), _ptr_array(hs),
                                         w_out.data_ptr(), _ptr(b_out), _ptr_array(addends) if addends else None, len(addends),
                                         label.data_ptr(), p_eps, pred.data_ptr(), partial.data_ptr(), M, _stream()),
               "rp_mlp_tail_fwd_bce")

    def finish():
        _check(lib().rp_loss_finish(partial.data_ptr(), n_part, weight / M, loss.data_ptr(), _stream()), "rp_loss_finish")

    if LaunchPlan.is_recording() and os.environ.get("RP_TAIL_REDUCE_SIDE", "1") != "0":
        LaunchPlan.defer_side(finish, (partial, loss))
    else:
        finish()
    return pred, loss, hs


def mlp_tail_bwd(dz, Ws, acts, w_out, bce=None):
    """-> (dhin [M,64] masked by hin > 0, [dW_l], [db_l], dw_out [1,64], db_out [1]) (rp_mlp_tail_bwd).
    bce = (pred, label, gloss, p_eps, weight, dz_out): the loss head's backward inside the launch (rp_mlp_tail_bwd_bce) —
    `dz` is then None and the logit's gradient is written to dz_out [M] for the other logit addends."""
    M, L = (dz.shape[0] if bce is None else bce[0].shape[0]), len(Ws)
    dev = dz.device if bce is None else bce[0].device
    dhin = torch.empty((M, 64), dtype=torch.float32, device=dev)
    grads = torch.empty((L * 4096 + L * 64 + 65,), dtype=torch.float32, device=dev)
    nbytes = _sz(0)
    _check(lib().rp_mlp_tail_bwd_workspace_bytes(M, L, C.byref(nbytes)), "rp_mlp_tail_bwd_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=dev)
    def launch(parts):
        if bce is not None:
            pred, label, gloss, p_eps, weight, dz_out = bce
            _check(lib().rp_mlp_tail_bwd_bce(pred.data_ptr(), label.data_ptr(), gloss.data_ptr(), p_eps, weight, _ptr(dz_out), L,
                                             _ptr_array(Ws), _i64_array([_rowmajor(w, "W") for w in Ws]), _ptr_array(acts),
                                             _rowmajor(acts[0], "hin"), w_out.data_ptr(), dhin.data_ptr(), 64, grads.data_ptr(),
                                             M, ws.data_ptr(), nbytes.value, parts, _stream()), "rp_mlp_tail_bwd_bce")
            return
        _check(lib().rp_mlp_tail_bwd_parts(dz.data_ptr(), L, _ptr_array(Ws), _i64_array([_rowmajor(w, "W") for w in Ws]),
                                           _ptr_array(acts), _rowmajor(acts[0], "hin"), w_out.data_ptr(), dhin.data_ptr(), 64,
                                           grads.data_ptr(), M, ws.data_ptr(), nbytes.value, parts, _stream()),
               "rp_mlp_tail_bwd")

    with _Timed("mlp_tail_bwd", f"{M}x64x{L}", 4 * M * (64 * (L + 2) + 1), 2 * M * (2 * 64 * 64 * L + 128)):
        if LaunchPlan.is_recording() and os.environ.get("RP_TAIL_REDUCE_SIDE", "1") != "0":
            # a captured step: the second stage (workspace -> grads, ~20 us of latency) reads nothing the following
            # launches write and nobody needs `grads` before the optimizer: it joins the plan's inline section (the second
            # side stream) behind the next launches recorded there, instead of standing between this launch and the first
            # layer's backward.  The workspace stays referenced until that section is joined.
            launch(1)
            LaunchPlan.defer_side(lambda: launch(2), (ws, grads))
        else:
            launch(3)
    dWs = [grads[l * 4096:(l + 1) * 4096].view(64, 64) for l in range(L)]
    dbs = [grads[L * 4096 + l * 64:L * 4096 + (l + 1) * 64] for l in range(L)]
    return dhin, dWs, dbs, grads[L * 4160:L * 4160 + 64].view(1, 64), grads[L * 4160 + 64:L * 4160 + 65]


_drop_calls = 0
# the dropout state of the GraphedTrainStep whose capture is running: {"seed", "clock": int64[1] device tensor, "calls"}
DROPOUT_CAPTURE = [None]


def _dropout_seed_offset(device):
    """(seed, offset) for one dropout call, tied to torch's RNG state: the seed is the device generator's (set by
    torch.manual_seed), the offset its Philox offset, advanced by 4 per call (the granularity torch accepts) so that
    other torch random ops interleave consistently.  Falls back to a process-wide call counter if this torch build does
    not expose generator offsets."""
    global _drop_calls
    if torch.cuda.is_current_stream_capturing():
        st = DROPOUT_CAPTURE[0]
        if st is None:
            raise RuntimeError("dropout inside a stream capture that is not a GraphedTrainStep's: its (seed, offset) are launch "
                               "arguments and would be frozen (every replay would draw the same mask)")
        # a captured step: the offset is read on the device (st['clock'] = the generator offset of the step's start) plus
        # this call's place in the step; the step's last launch advances the clock (graph_step.GraphedTrainStep)
        delta = 4 * st["calls"]
        st["calls"] += 1
        return st["seed"], delta, st["clock"]
    try:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
        return seed & 0xFFFFFFFFFFFFFFFF, off, None
    except (AttributeError, RuntimeError):
        _drop_calls += 1
        return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, 4 * _drop_calls, None


def dropout_fwd(x, p: float, seed: Optional[int] = None, offset: Optional[int] = None):
    """-> (y, mask uint8 [M, N]); seed / offset default to torch's device generator state (advanced)."""
    _req(x, torch.float32, "x")
    M, N = x.shape
    clock = None
    if seed is None:
        seed, offset, clock = _dropout_seed_offset(x.device)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    mask = torch.empty((M, N), dtype=torch.uint8, device=x.device)
    with _Timed("dropout_fwd", f"{M}x{N}", 9 * M * N):
        if clock is not None:
            _check(lib().rp_dropout_fwd_dev(x.data_ptr(), _rowmajor(x, "x"), y.data_ptr(), N, mask.data_ptr(), M, N, p, seed,
                                            offset, clock.data_ptr(), _stream()), "rp_dropout_fwd_dev")
        else:
            _check(lib().rp_dropout_fwd(x.data_ptr(), _rowmajor(x, "x"), y.data_ptr(), N, mask.data_ptr(), M, N, p, seed, offset,
                                        _stream()), "rp_dropout_fwd")
    return y, mask


def counter_add_u64(counter, delta: int):
    """*counter += delta (uint64 / int64 device scalar) on the current stream (rp_counter_add_u64)"""
    _check(lib().rp_counter_add_u64(counter.data_ptr(), int(delta), _stream()), "rp_counter_add_u64")


def device_generator(device):
    """torch's default generator of a HIP device"""
    return torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]


def dropout_bwd(dy, mask, p: float):
    M, N = dy.shape
    dx = torch.empty((M, N), dtype=torch.float32, device=dy.device)
    with _Timed("dropout_bwd", f"{M}x{N}", 9 * M * N):
        _check(lib().rp_dropout_bwd(dy.data_ptr(), _rowmajor(dy, "dy"), mask.data_ptr(), dx.data_ptr(), N, M, N, p, _stream()),
               "rp_dropout_bwd")
    return dx


def adam_step_scalars(lr: float, beta1: float, beta2: float, step: int, eps: float = 1e-8):
    """the two per-step scalars of the Adam kernels (see rp_adam_step_scalars): one row of the lazy step table"""
    a, b = C.c_float(0), C.c_float(0)
    _check(lib().rp_adam_step_scalars(lr, beta1, beta2, eps, step, C.byref(a), C.byref(b)), "rp_adam_step_scalars")
    return a.value, b.value


def adam_step_scalars_range(lr: float, beta1: float, beta2: float, step0: int, n: int, eps: float = 1e-8) -> torch.Tensor:
    """[n, 2] float32 HOST tensor: row i = adam_step_scalars(.., step0 + i) — one C call (rp_adam_step_scalars_range)"""
    out = torch.empty((n, 2), dtype=torch.float32)
    _check(lib().rp_adam_step_scalars_range(lr, beta1, beta2, eps, step0, n, out.data_ptr()), "rp_adam_step_scalars_range")
    return out


def lazy_adam_rows(sorted_keys, D: int, p, g, m, v, last, scalars, t_target: int, real_step: bool, zero_grad: bool,
                   beta1: float, beta2: float, eps: float, cf_table=None, cf_from: int = 0, t_dev=None):
    """cf_table (from lazy_adam_cf_table, built for the end step of this call's replay) selects the closed-form replay
    of the steps after `cf_from`; None = the bit-exact serial replay"""
    with _Timed("lazy_adam_rows_step" if real_step else "lazy_adam_rows_replay", f"D={D}"):
        _check(lib().rp_lazy_adam_rows(sorted_keys.data_ptr(), sorted_keys.numel(), D, p.data_ptr(), _ptr(g),
                                       m.data_ptr(), v.data_ptr(), last.data_ptr(), scalars.data_ptr(), t_target,
                                       int(real_step), int(zero_grad), beta1, beta2, eps, _ptr(cf_table), cf_from,
                                       _ptr(t_dev), _stream()),
               "rp_lazy_adam_rows")


def lazy_adam_flush(rows: int, D: int, p, m, v, last, scalars, t_target: int, beta1: float, beta2: float, eps: float,
                    cf_table=None, cf_from: int = 0):
    with _Timed("lazy_adam_flush", f"D={D}"):
        _check(lib().rp_lazy_adam_flush(rows, D, p.data_ptr(), m.data_ptr(), v.data_ptr(), last.data_ptr(),
                                        scalars.data_ptr(), t_target, beta1, beta2, eps, _ptr(cf_table), cf_from,
                                        _stream()),
               "rp_lazy_adam_flush")


def lazy_adam_catchup(sorted_keys, D: int, p, g, m, v, last, scalars, t_done: int, mark, beta1: float,
                      beta2: float, eps: float, cf_table=None, cf_from: int = 0, t_dev=None, shadow=None):
    """deferred execution (rp_lazy_adam_catchup): everything the unique rows of sorted_keys are owed through step t_done —
    their pending real step, then the zero-gradient steps — and, with mark, the stamp 'gradient of step t_done+1 coming'.
    mark = 2: stamp, and leave the applied gradient rows uncleared (the caller's backward overwrites them).
    shadow: the bf16 lookup copy of the tables (bf16-storage training), written wherever a parameter row is"""
    with _Timed("lazy_adam_catchup", f"D={D}"):
        _check(lib().rp_lazy_adam_catchup(sorted_keys.data_ptr(), sorted_keys.numel(), D, p.data_ptr(), _ptr(g),
                                          m.data_ptr(), v.data_ptr(), last.data_ptr(), scalars.data_ptr(), t_done,
                                          int(mark), beta1, beta2, eps, _ptr(cf_table), cf_from, _ptr(t_dev), _ptr(shadow),
                                          _stream()), "rp_lazy_adam_catchup")


def lazy_adam_flush_deferred(rows: int, D: int, p, g, m, v, last, scalars, t_target: int, beta1: float, beta2: float,
                             eps: float, cf_table=None, cf_from: int = 0, shadow=None):
    with _Timed("lazy_adam_flush", f"D={D}"):
        _check(lib().rp_lazy_adam_flush_deferred(rows, D, p.data_ptr(), _ptr(g), m.data_ptr(), v.data_ptr(),
                                                 last.data_ptr(), scalars.data_ptr(), t_target, beta1, beta2, eps,
                                                 _ptr(cf_table), cf_from, _ptr(shadow), _stream()),
               "rp_lazy_adam_flush_deferred")


def rows_to_bf16(sorted_keys, D: int, p, shadow):
    """shadow[row] = bf16(p[row]) for the unique rows of sorted_keys (rp_rows_to_bf16)"""
    _req(shadow, torch.bfloat16, "shadow")
    with _Timed("rows_to_bf16"):
        _check(lib().rp_rows_to_bf16(sorted_keys.data_ptr(), sorted_keys.numel(), D, p.data_ptr(), shadow.data_ptr(), _stream()),
               "rp_rows_to_bf16")


def lazy_adam_cf_table(ns_d, t_end: int, cf_from: int, beta1: float, beta2: float, cf_table, built_to: int = -1, t_dev=None):
    """bring the closed-form replay table to replays ending at step t_end (see rp_lazy_adam_cf_table): ns_d [>= t_end + 1, 2]
    float64 device table by step, cf_table [capacity > t_end, 8] float32 (written in place).  built_to: the t_end of the
    previous call on this buffer (only the stamps that were not final then are rebuilt), -1 = a fresh buffer."""
    assert ns_d.dtype == torch.float64 and ns_d.is_contiguous() and (t_dev is not None or ns_d.shape[0] > t_end)
    assert cf_table.dtype == torch.float32 and cf_table.is_contiguous() and (t_dev is not None or cf_table.shape[0] > t_end)
    with _Timed("lazy_adam_cf_table"):
        _check(lib().rp_lazy_adam_cf_table(ns_d.data_ptr(), t_end, cf_from, beta1, beta2, cf_table.data_ptr(), cf_table.shape[0],
                                           built_to, _ptr(t_dev), _stream()), "rp_lazy_adam_cf_table")


POOL_MODES = {"sum": 0, "average": 1}


def embed_gather_pool_fwd(arena, row_base: int, row_count: int, ids, offsets, L: int, B: int, mode: str, err_flag,
                          want_bwd: bool):
    """rp_embed_gather_pool_fwd: pooled multi-id lookup of ONE table.  ids: flat int64 [nnz] (dense bags: nnz = B * L,
    offsets None; CSR: offsets int64 [B + 1]).  Returns (out [B, D], inv [B, D] or None, bag_of int32 [nnz] or None)."""
    _req(arena, torch.float32, "arena")
    D = arena.shape[1]
    out = torch.empty((B, D), dtype=torch.float32, device=arena.device)
    inv = torch.empty((B, D), dtype=torch.float32, device=arena.device) if (want_bwd and mode == "average") else None
    bag = torch.empty((ids.numel(),), dtype=torch.int32, device=arena.device) if (want_bwd and offsets is not None) else None
    nnz = ids.numel()
    with _Timed("embed_gather_pool_fwd", f"D={D}", nnz * (D * 4 + 8) + B * D * 4):
        _check(lib().rp_embed_gather_pool_fwd(arena.data_ptr(), row_base, row_count, ids.data_ptr(), _ptr(offsets), L, B, D,
                                              POOL_MODES[mode], out.data_ptr(), D, _ptr(inv), _ptr(bag), err_flag.data_ptr(),
                                              _stream()), "rp_embed_gather_pool_fwd")
    return out, inv, bag


def embed_pool_bwd(sorted_keys, sorted_pos, D: int, g, scale, bag_of, L: int, grad_arena, accumulate: bool):
    _req(grad_arena, torch.float32, "grad_arena")
    _req(g, torch.float32, "g")
    n = sorted_keys.numel()
    nbytes = _sz(0)
    _check(lib().rp_embed_grad_reduce_workspace_bytes(n, D, C.byref(nbytes)), "rp_embed_grad_reduce_workspace_bytes")
    ws = torch.empty((nbytes.value,), dtype=torch.uint8, device=grad_arena.device)
    with _Timed("embed_pool_bwd", f"D={D}", n * (D * 4 + 8)):
        _check(lib().rp_embed_pool_bwd(sorted_keys.data_ptr(), sorted_pos.data_ptr(), n, D, g.data_ptr(), _rowmajor(g, "g"),
                                       _ptr(scale), _ptr(bag_of), L, grad_arena.data_ptr(), int(accumulate), ws.data_ptr(),
                                       nbytes.value, _stream()), "rp_embed_pool_bwd")


def seq_pool_fwd(e, mode: str, want_bwd: bool):
    """MaskedSumPooling / MaskedAveragePooling of an explicit contiguous [B, L, D] tensor -> (out [B, D], inv or None)"""
    _req(e, torch.float32, "e")
    B, L, D = e.shape
    out = torch.empty((B, D), dtype=torch.float32, device=e.device)
    inv = torch.empty((B, D), dtype=torch.float32, device=e.device) if (want_bwd and mode == "average") else None
    with _Timed("seq_pool_fwd", f"{B}x{L}x{D}", (B * L * D + B * D) * 4):
        _check(lib().rp_seq_pool_fwd(e.data_ptr(), B, L, D, POOL_MODES[mode], out.data_ptr(), _ptr(inv), _stream()),
               "rp_seq_pool_fwd")
    return out, inv


def seq_pool_bwd(g, inv, L: int):
    _req(g, torch.float32, "g")
    B, D = g.shape
    de = torch.empty((B, L, D), dtype=torch.float32, device=g.device)
    with _Timed("seq_pool_bwd", f"{B}x{L}x{D}", (B * L * D + B * D) * 4):
        _check(lib().rp_seq_pool_bwd(g.data_ptr(), _ptr(inv), B, L, D, de.data_ptr(), _stream()), "rp_seq_pool_bwd")
    return de
