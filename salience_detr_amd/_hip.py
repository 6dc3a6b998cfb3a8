"""ctypes binding of libsalience_hip.so (the C ABI declared in include/salience_hip.h).

The product path has NO CPU fallback: if the library is missing, or an operator is handed a
tensor that is not on a HIP device, the call raises.  PyTorch is used here only as the owner
of device memory and streams (``tensor.data_ptr()``, ``torch.cuda.current_stream()``).
"""
import ctypes
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsalience_hip.so")
# The fp16-activation flavour (round 5): the same sources built with -DSDETR_ACT_F16, same C ABI, every 16-bit ACTIVATION
# (token rows, projection slabs, 16-bit outputs, packed weights) IEEE half instead of bfloat16 (csrc/common.h).  An
# operator picks the library by the dtype of the activations it is handed: ``lib(x.dtype)``.
F16_LIB_PATH = os.path.join(_HERE, "libsalience_hip_f16.so")

F32, BF16, F16 = 0, 1, 2
EINVAL = -1
ACT16 = (torch.bfloat16, torch.float16)      # the two 16-bit activation types (one library each)

_lib = None
_lib_f16 = None
_tls = threading.local()   # .last = the library of this thread's most recent lib() call: where check() reads the error text


def is_act16(dt) -> bool:
    return dt in ACT16

_i = ctypes.c_int


class FinalizeJobStruct(ctypes.Structure):
    """``sdetr_finalize_job`` of include/salience_hip.h."""
    _fields_ = [("tokens", ctypes.c_void_p), ("background", ctypes.c_void_p), ("padding_mask", ctypes.c_void_p),
                ("batch", ctypes.c_int), ("spatial_size", ctypes.c_int), ("out", ctypes.c_void_p)]


class BorderedLayoutStruct(ctypes.Structure):
    """``sdetr_bordered_layout`` of include/salience_hip.h."""
    _fields_ = [("pixel_map", ctypes.c_void_p), ("border", ctypes.c_void_p), ("num_border", ctypes.c_int),
                ("records", ctypes.c_int)]


class RowOrdersJobStruct(ctypes.Structure):
    """``sdetr_row_orders_job`` of include/salience_hip.h."""
    _fields_ = [("sorted_index", ctypes.c_void_p), ("index_batch_stride", ctypes.c_int64), ("tile_pos", ctypes.c_void_p),
                ("batch", ctypes.c_int), ("spatial_size", ctypes.c_int), ("num_rows", ctypes.c_int),
                ("num_layers", ctypes.c_int), ("counts", ctypes.c_void_p), ("order", ctypes.c_void_p),
                ("order_batch_stride", ctypes.c_int64)]


class RankJobStruct(ctypes.Structure):
    """``sdetr_rank_job`` of include/salience_hip.h."""
    _fields_ = [("score", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("mask_row_stride", ctypes.c_int64),
                ("fill_value", ctypes.c_void_p), ("batch", ctypes.c_int), ("n", ctypes.c_int), ("k", ctypes.c_int),
                ("index_offset", ctypes.c_int64), ("out_score", ctypes.c_void_p), ("out_index", ctypes.c_void_p),
                ("out_row_stride", ctypes.c_int64)]
_i64 = ctypes.c_int64
_p = ctypes.c_void_p
_sz = ctypes.c_size_t

# name -> (restype, argtypes); must match include/salience_hip.h
SIGNATURES = {
    "sdetr_abi_version": (_i, []),
    "sdetr_last_error": (ctypes.c_char_p, []),
    "sdetr_msda_im2col_f32": (_i, [_p] * 6 + [_i] * 7 + [_p]),
    "sdetr_msda_im2col_f64": (_i, [_p] * 6 + [_i] * 7 + [_p]),
    "sdetr_msda_col2im_f32": (_i, [_p] * 7 + [_i] * 7 + [_p] * 3),
    "sdetr_msda_col2im_f64": (_i, [_p] * 7 + [_i] * 7 + [_p] * 3),
    "sdetr_msda_col2im_lds_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "sdetr_msda_col2im_lds_supported": (_i, [_i] * 5),
    "sdetr_msda_col2im_lds_f32": (_i, [_p] * 7 + [_i] * 7 + [_p] * 3 + [_p, _sz]),
    "sdetr_msda_last_backward_kernel": (_i, []),
    "sdetr_value_to_head_major": (_i, [_p, _p, _i, _i64, _p, _i, _i, _i, _i, _i, _p, _i]),
    "sdetr_msda_fused_forward": (_i, [_p, _p, _i, _p, _p, _p, _i, _i64, _p, _i, _i64, _i, _p] + [_i] * 7 + [_p, _i]),
    "sdetr_msda_forward_head_major": (_i, [_p, _p, _i, _p, _p, _p, _p] + [_i] * 7 + [_p, _i]),
    "sdetr_msda_resident_max_pixels": (_i, []),
    "sdetr_msda_resident_forward": (_i, [_p, _p, _i, _p, _p, _i, _i64, _p, _i, _i, _i, _i, _p, _i, _i]),
    "sdetr_msda_last_kernel": (_i, []),
    "sdetr_msda_bordered_records": (_i64, [_p, _i]),
    "sdetr_layer_row_orders": (_i, [_p, _p, _i64, _p, _i, _i, _i, _i, _p, _p, _i64]),
    "sdetr_msda_bordered_max_resident_records": (_i, []),
    "sdetr_msda_bordered_forward": (_i, [_p, _p, _i, _p, _p, _i, _i64, _p, _p, _i64, _i, _i, _i, _i, _p, _i, _i]),
    "sdetr_msda_bordered_forward_ex": (_i, [_p, _p, _i, _p, _p, _i, _i64, _p, _p, _i64, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i]),
    "sdetr_msda_resident_forward_ex": (_i, [_p, _p, _i, _p, _p, _i, _i64, _p, _i, _i, _i, _i, _p, _i, _i, _i]),
    "sdetr_gemm_x3_generation": (_i, [_i]),
    "sdetr_topk_attention_workspace_bytes": (_i64, [_i, _i]),
    "sdetr_topk_attention_bf16": (_i, [_p, _p, _i64, _p, _i64, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, ctypes.c_float, _i, _i,
                                       _p, _i64]),
    "sdetr_topk_workspace_bytes": (_sz, [_i, _i, _i]),
    "sdetr_topk_uses_prefilter": (_i, [_i, _i]),
    "sdetr_masked_topk_desc_f32": (_i, [_p, _p, _p, _i64, _i, _p, _p, _i, _i, _i, _i64, _p, _p, _i64, _p, _sz]),
    "sdetr_masked_topk_desc_with_orders_f32": (_i, [_p, _p, _p, _i64, _i, _p, _p, _i, _i, _i, _i64, _p, _p, _i64, _p, _sz, _p]),
    "sdetr_topk_sliced_workspace_bytes": (_sz, [_i, _i]),
    "sdetr_masked_topk_sliced_f32": (_i, [_p, _p, _p, _i64, _p, _i, _i, _i, _i, _i64, _p, _p, _i64, _p, _sz]),
    "sdetr_masked_topk_sliced_with_rank_f32": (_i, [_p, _p, _p, _i64, _p, _i, _i, _i, _i, _i64, _p, _p, _i64, _p, _sz, _p, _p, _p]),
    "sdetr_merge_sorted_desc": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "sdetr_attention_train_max_rows": (_i, []),
    "sdetr_attention_train_forward_f32": (_i, [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, ctypes.c_float,
                                                _p, _p]),
    "sdetr_attention_train_backward_f32": (_i, [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, ctypes.c_float,
                                                 _p, _p, _p, _p, _p, _p]),
    "sdetr_sampling_prep_supported": (_i, [_i, _i]),
    "sdetr_sampling_prep_f32": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p, _p]),
    "sdetr_sampling_prep_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p, _p]),
    "sdetr_layer_norm_train_supported": (_i, [_i]),
    "sdetr_layer_norm_train_forward_f32": (_i, [_p, _p, _p, _p, _p, ctypes.c_float, _i64, _i, _p, _p, _p, _p]),
    "sdetr_layer_norm_train_backward_f32": (_i, [_p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p]),
    "sdetr_masked_fill_min": (_i, [_p, _p, _p, _p, _i, _i64, _p]),
    "sdetr_decoder_query_sine_embed": (_i, [_p, _p, _p, _i, _i, _i, _i, ctypes.c_float, _p, _i, _p]),
    "sdetr_box_refine": (_i, [_p, _p, _i, _i64, _p, _i64, _i, ctypes.c_float, _p]),
    "sdetr_mlp_rows_bf16": (_i, [_p, _p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _i, _p, _i64]),
    "sdetr_rows_linear_bf16": (_i, [_p, _p, _p, _i64, _i, _p, _p, _i, _p, _i64]),
    "sdetr_rows_linear_ln_bf16": (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, ctypes.c_float, _p]),
    "sdetr_ref_point_head_bf16": (_i, [_p, _p, _p, _i, _i, _i, ctypes.c_float, _p, _p, _p, _p, _p, _p]),
    "sdetr_decoder_head_bf16": (_i, [_p, _p, _i64, _p, _p, ctypes.c_float, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p,
                                     ctypes.c_float, _i, _p, _i64, _p]),
    "sdetr_encoder_output_proposals": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "sdetr_grid_nms_topk": (_i, [_p, _p, _i64, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "sdetr_proposal_refine": (_i, [_p, _p, _i, _p, _p, _i64, _i, _i, _i, _p]),
    "sdetr_salience_targets": (_i, [_p, _p, _p, _i, _p, _p, _p, _i, ctypes.c_float, _p, _p]),
    "sdetr_focal_loss_workspace_bytes": (_i64, [_i64]),
    "sdetr_salience_focal_loss": (_i, [_p, _p, _p, _i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, _p, _i64, _p]),
    "sdetr_salience_focal_loss_backward": (_i, [_p, _p, _p, _i64, ctypes.c_float, ctypes.c_float, _p, _p, _p]),
    "sdetr_attention_heads_bf16": (_i, [_p, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i, _i, _i, _i, ctypes.c_float, _p]),
    "sdetr_encoder_prepare_sorted": (_i, [_p, _p, _p, _i, _p, _p, _i64, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i]),
    "sdetr_encoder_prepare_sorted_scored": (_i, [_p, _p, _p, _i, _p, _p, _i64, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i,
                                                  _p, _p, _p]),
    "sdetr_encoder_reference_points": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _p]),
    "sdetr_pyramid_flatten_level": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i]),
    "sdetr_pyramid_flatten": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "sdetr_class_max_times": (_i, [_p, _p, _i, _p, _i64, _i, _i, _i, _p]),
    "sdetr_layernorm": (_i, [_p, _p, _p, _i, _i64, _i64, _i64, _i64, _p, _p, _p, _p, _i, ctypes.c_float, _i, _i, _i, _p, _i,
                             _p, _i64, _i]),
    "sdetr_advance_rows": (_i, [_p, _p, _p, _p, _p, _p, _i64, _p, _i, _i, _i, _i, _i, _i]),
    "sdetr_select_stack": (_i, [_p, _p, _i64, _p, _i64, _p, _i, _i, _i, _i, _p]),
    "sdetr_encoder_finalize_sorted": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sdetr_encoder_finalize": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "sdetr_column_mean_f32": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p]),
    "sdetr_pack_linear_f32": (_i, [_p, _p, _i64, _i, _i, _p]),
    "sdetr_salience_head_blocks": (_i, [_i, _i]),
    "sdetr_salience_head_stage1": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _i, _i,
                                        _i, _i, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i64, _p, _p]),
    "sdetr_salience_head_stage1_x3": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _i, _i,
                                        _i, _i, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i64, _p, _p]),
    "sdetr_salience_head_const": (_i, [_p, _p, _i, _i, _p, _p, _p, _p]),
    "sdetr_stage2_with_value_proj": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p,
                                          _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    "sdetr_stage1_x3_with_value_proj": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _i, _i,
                                        _i, _i, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i64, _p, _p,
                                             _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p]),
    "sdetr_salience_head_hoist_x3": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i64, _p,
                                          _i64, _p, _i64, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p]),
    "sdetr_salience_head_modulate": (_i, [_p, _p, _i64, _p, _i64, _i, _i, _p, _p, _i, _i, _i, _i, _p, ctypes.c_float, _p, _p, _p,
                                          _p, _p, _p]),
    "sdetr_stage1_x3_with_jobs": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _i, _i,
                                        _i, _i, _p, _p, _p, ctypes.c_float, _p, _p, _p, _i64, _p, _p,
                                             _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _p, _p]),
    "sdetr_pack_linear_bf16x3": (_i, [_p, _p, _i64, _i, _i, _p]),
    "sdetr_salience_head_stage2": (_i, [_p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p, _p, _i]),
    "sdetr_ffn_packed_bytes": (_i64, [_i]),
    "sdetr_ffn_pack_bf16": (_i, [_p, _p, _p, _i, _i, _p]),
    "sdetr_ffn_auto_splits": (_i, [_i, _i]),
    "sdetr_ffn_workspace_bytes": (_i64, [_i, _i]),
    "sdetr_ffn_fused_advance_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, ctypes.c_float, _i, _i, _i, _i, _i, _p, _i64, _p, _p, _p, _p,
                                          _i64, _p, _i, _i, _i]),
    "sdetr_attn_tail_packed_bytes": (_i64, []),
    "sdetr_attn_tail_pack_bf16": (_i, [_p, _p, _i, _p]),
    "sdetr_attn_tail_ffn_advance_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, ctypes.c_float, _p, _p, _p, _p, ctypes.c_float, _i, _i,
                                              _i, _i, _i, _p, _i64, _p, _p, _p, _p, _i64, _p, _i, _i, _i, _p, _p, _i64, _p]),
    "sdetr_class_head_packed_bytes": (_i64, []),
    "sdetr_class_head_pack_bf16": (_i, [_p, _p, _i, _i, _p]),
    "sdetr_gemm_x3_presplit": (_i, [_p, _p, _i64, _i, _i, _i, _p]),
    "sdetr_gemm_x3_f32": (_i, [_p, _p, _i64, _i, _p, _i64, _i, _p, _i64, _i, _i, _i, _p, _i, _p]),
    "sdetr_gemm_x3_epilogue_f32": (_i, [_p, _p, _i64, _i, _p, _i64, _i, _p, _i64, _i, _i, _i, _p, _i, _p, _i, _p, _i64]),
    "sdetr_topk_attention_with_projection_bf16": (_i, [_p, _p, _i64, _p, _i64, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p,
                                                       ctypes.c_float, _p, _i64, _p, _p, _p, _p, _p, _i64, _p, _p, _i]),
    "sdetr_topk_select_inproj_bf16": (_i, [_p, _p, _i, _i, _i, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _p, _i64, _p, _p, _i64]),
    "sdetr_topk_select_candidate_bytes": (_i64, [_i, _i, _i]),
    "sdetr_topk_inproj_launch": (_i, [_p, _p]),
    "sdetr_ffn_fused_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, ctypes.c_float, _i, _i, _i, _p, _i, _p, _i64]),
    "sdetr_linear_packed_bytes": (_i64, [_i]),
    "sdetr_linear_pack_bf16": (_i, [_p, _p, _i64, _i, _i, _p]),
    "sdetr_token_linear_bf16": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p, _i, _p, _i64, _i]),
    "sdetr_value_proj_head_major": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "sdetr_class_head_max_times": (_i, [_p, _p, _p, _p, _i, _i, _p, _i64, _i, _i, _p]),
    "sdetr_token_linear_ln_bf16": (_i, [_p, _p, _p, _i64, _i, _i, _i, _p, _p, _p, _p, ctypes.c_float, _p, _p, _i64]),
    "sdetr_gather_rows": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "sdetr_scatter_rows": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i]),
    "sdetr_neck_conv3x3": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "sdetr_neck_conv3x3_packed_bytes": (_i64, [_i, _i, _i]),
    "sdetr_neck_pack_conv3x3_bf16": (_i, [_p, _p, _i, _i, _i, _p]),
    "sdetr_neck_conv3x3_mfma_bf16": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "sdetr_neck_combine": (_i, [_p, _p, _i, _p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i]),
    "sdetr_neck_gate_workspace_bytes": (_i64, [_i, _i, _i]),
    "sdetr_neck_gate_shortcut": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _i, _p, _i64, _p, _p]),
}


class HipExtensionError(RuntimeError):
    pass


def _load(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise HipExtensionError(
            f"{path} is missing: build it with `python -m salience_detr_amd.csrc.build` "
            "(or __graft_entry__.build()); there is no CPU fallback for the hot path")
    cdll = ctypes.CDLL(path)          # RTLD_LOCAL: the two flavours export the same names and do not see each other
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError -> the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if cdll.sdetr_abi_version() != 1:
        raise HipExtensionError(f"{os.path.basename(path)} ABI version mismatch")
    return cdll


def lib(act=None) -> ctypes.CDLL:
    """Load the shared library once; fail loudly if it is not built.  ``act``: the dtype (or a tensor) of the 16-bit
    activations of the call -- ``torch.float16`` selects the fp16-activation flavour, anything else the bf16 library (which
    also holds every fp32 / integer operator)."""
    global _lib, _lib_f16
    if act is not None and not isinstance(act, torch.dtype):
        act = act.dtype
    if act == torch.float16:
        if _lib_f16 is None:
            _lib_f16 = _load(F16_LIB_PATH)
        _tls.last = _lib_f16
        return _lib_f16
    if _lib is None:
        _lib = _load(LIB_PATH)
    _tls.last = _lib
    return _lib


def check(code: int, what: str, library: Optional[ctypes.CDLL] = None) -> None:
    """Raises on a non-zero status.  The error text is thread-local inside each library; ``library`` names the one the
    failing call went to (default: the library THIS thread's most recent ``lib()`` call selected -- ADVICE r5)."""
    if code != 0:
        msg = (library or getattr(_tls, "last", None) or lib()).sdetr_last_error().decode(errors="replace")
        if code == EINVAL:
            raise RuntimeError(f"{what}: {msg}")
        raise RuntimeError(f"{what}: HIP launch error {code}: {msg}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_device(what: str, **tensors) -> None:
    """The reference asserts `.is_cuda` + contiguity (ms_deform_attn_cuda.cu:20-30); so do we."""
    for name, t in tensors.items():
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(f"{what}: {name} must be a HIP (cuda) tensor; the hot path has no CPU fallback")
        if not t.is_contiguous():
            raise RuntimeError(f"{what}: {name} tensor has to be contiguous")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16  # head-major value maps (either library); activations of the fp16 flavour (lib(torch.float16))
    raise RuntimeError(f"unsupported dtype {dt} (float32 / bfloat16 / float16)")
