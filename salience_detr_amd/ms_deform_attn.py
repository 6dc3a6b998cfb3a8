"""Multi-scale deformable attention on MI355X behind the reference's operator signatures.

Mirrors (names, argument meaning, error behaviour) of the reference's
``models/bricks/ms_deform_attn.py``:

* B1  ``ms_deform_attn_forward`` / ``ms_deform_attn_backward`` -- the two functions the
  reference's pybind module ``_C`` exports (``models/bricks/ops/cuda/ms_deform_attn_cuda.cu:12-18,
  75-82, 148-151``), here thin host wrappers over the C ABI ``sdetr_msda_im2col_f32`` /
  ``sdetr_msda_col2im_f32`` of libsalience_hip.so.
* B2  ``MultiScaleDeformableAttnFunction`` (``ms_deform_attn.py:35-84``).
* B3  ``MultiScaleDeformableAttention`` (``ms_deform_attn.py:215-377``): same constructor, same
  parameter names (``sampling_offsets``, ``attention_weights``, ``value_proj``, ``output_proj`` --
  released checkpoints and ``optimizer/param_dict.py:79-81`` depend on them), same ``forward``
  signature.  Without autograd it runs the native path: value re-laid head-major (optionally
  bf16) and ONE fused kernel for softmax + sampling locations + gather.

There is no CPU fallback (the reference's ``multi_scale_deformable_attn_pytorch`` lives, restated,
in ``oracle/`` as the parity checker only).
"""
import math
import warnings
from typing import Optional

import threading

import torch
from torch import Tensor, nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn import functional as F
from torch.nn.init import constant_, xavier_uniform_

from . import _hip


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


def _op_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, what):
    _hip.require_device(what, value=value, spatial_shapes=spatial_shapes, level_start_index=level_start_index,
                        sampling_loc=sampling_loc, attn_weight=attn_weight)
    if value.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"{what}: only float32 / float64 are supported by the reference-layout op "
                           f"(got {value.dtype})")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError(f"{what}: value, sampling_loc and attn_weight must share one dtype")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError(f"{what}: spatial_shapes / level_start_index must be int64")
    B, Nv, M, D = value.shape
    L = spatial_shapes.shape[0]
    Nq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (B, Nq, M, L, P, 2) or tuple(attn_weight.shape) != (B, Nq, M, L, P):
        raise RuntimeError(f"{what}: inconsistent shapes value{tuple(value.shape)} "
                           f"loc{tuple(sampling_loc.shape)} aw{tuple(attn_weight.shape)}")
    step = min(B, int(im2col_step)) if B > 0 else 1
    if step <= 0 or B % step != 0:
        # reference: AT_ASSERTM(batch % im2col_step_ == 0, ...)  ms_deform_attn_cuda.cu:42-44
        raise RuntimeError(f"{what}: batch({B}) must divide im2col_step({step})")
    return B, Nv, M, D, L, Nq, P


def ms_deform_attn_forward(value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                           sampling_loc: Tensor, attn_weight: Tensor, im2col_step: int) -> Tensor:
    """``_C.ms_deform_attn_forward`` (ms_deform_attn_cuda.cu:12-72).  Returns ``[B, Nq, M*D]``.

    The reference chunks the batch by ``im2col_step`` only to bound its int32 thread index; the
    HIP launcher covers the whole batch in one launch, so the step is validated and otherwise unused.
    """
    B, Nv, M, D, L, Nq, P = _op_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                     im2col_step, "ms_deform_attn_forward")
    out = torch.empty((B, Nq, M * D), dtype=value.dtype, device=value.device)
    fn = _hip.lib().sdetr_msda_im2col_f32 if value.dtype == torch.float32 else _hip.lib().sdetr_msda_im2col_f64
    with torch.cuda.device(value.device):
        code = fn(_hip.stream_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                  sampling_loc.data_ptr(), attn_weight.data_ptr(), B, Nv, M, D, L, Nq, P, out.data_ptr())
    _hip.check(code, "ms_deform_attn_forward")
    return out


# The LDS-accumulating backward pays three launches and a flush of every window (~100 us at the benchmark pyramid,
# whatever the query count): measured on MI355X (benchmarks/msda_backward_ab.py) it wins from ~1200 queries per
# image up (3.1x at 11 363), the direct kernel (one launch, global atomics) below.  `lds_backward = False` forces
# the direct kernel.
lds_backward = True
lds_backward_min_queries = 1200


def last_backward_kernel() -> int:
    """``SDETR_KERNEL_MSDA_BWD_*`` code of the kernel the calling thread's last MSDA backward call dispatched to."""
    return int(_hip.lib().sdetr_msda_last_backward_kernel())


KERNEL_BWD_DIRECT, KERNEL_BWD_LDS = 1, 2


def ms_deform_attn_backward(value: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                            sampling_loc: Tensor, attn_weight: Tensor, grad_output: Tensor, im2col_step: int):
    """``_C.ms_deform_attn_backward`` (ms_deform_attn_cuda.cu:75-145).
    Returns ``[grad_value, grad_sampling_loc, grad_attn_weight]``."""
    B, Nv, M, D, L, Nq, P = _op_dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                     im2col_step, "ms_deform_attn_backward")
    _hip.require_device("ms_deform_attn_backward", grad_output=grad_output)
    if grad_output.dtype != value.dtype or grad_output.numel() != B * Nq * M * D:
        raise RuntimeError("ms_deform_attn_backward: grad_output must be [B, Nq, M*D] of value's dtype")
    from . import zero_arena
    grad_value = zero_arena.zeros_like(value)   # (the op accumulates into it; inside a ZeroArena step: a slice of the step's one fill)
    grad_loc = torch.empty_like(sampling_loc)
    grad_aw = torch.empty_like(attn_weight)
    lib = _hip.lib()
    common = (grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
              sampling_loc.data_ptr(), attn_weight.data_ptr(), B, Nv, M, D, L, Nq, P, grad_value.data_ptr(),
              grad_loc.data_ptr(), grad_aw.data_ptr())
    with torch.cuda.device(value.device):
        if (lds_backward and value.dtype == torch.float32 and Nq >= lds_backward_min_queries
                and lib.sdetr_msda_col2im_lds_supported(M, D, L, P, Nv)):
            # grad_value accumulated in LDS windows (msda_backward_tiled.hip); scratch for the query bucketing
            nbytes = lib.sdetr_msda_col2im_lds_workspace_bytes(B, Nq, M, L)
            workspace = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
            code = lib.sdetr_msda_col2im_lds_f32(_hip.stream_ptr(), *common, workspace.data_ptr(), nbytes)
        else:
            fn = lib.sdetr_msda_col2im_f32 if value.dtype == torch.float32 else lib.sdetr_msda_col2im_f64
            code = fn(_hip.stream_ptr(), *common)
    _hip.check(code, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_aw]


class MultiScaleDeformableAttnFunction(Function):
    """Autograd wrapper with the reference's argument order (ms_deform_attn.py:35-84)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                        sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        grad_value, grad_loc, grad_aw = ms_deform_attn_backward(
            value, shapes, lsi, loc, aw, grad_output.contiguous(), ctx.im2col_step)
        # grad_value was allocated by this call and nothing else holds it yet: marked as exclusively owned, so that the
        # padding mask behind the op (_ZeroRowsInPlace) may zero its rows in place (the consumer used to guess
        # this from version counters)
        _mark_exclusive(grad_value)
        return grad_value, None, None, grad_loc, grad_aw, None


# ------------------------------------------------------------------------------------------------
# native-layout helpers (no reference symbol; the inside of MultiScaleDeformableAttention.forward)
# ------------------------------------------------------------------------------------------------
def value_to_head_major(value_proj_out: Tensor, key_padding_mask: Optional[Tensor], num_heads: int,
                        out_dtype: Optional[torch.dtype] = None, num_groups: int = 1) -> Tensor:
    """masked_fill(padding, 0) + head split of ms_deform_attn.py:318-321, written ``[B, M, Nv, D]``.

    ``value_proj_out`` is ``[B, Nv, G*M*D]`` and may be a column slice of a wider GEMM output (row stride
    > G*M*D).  With ``num_groups`` G > 1 (the batched value projection of all encoder layers) the result is
    ``[G, B, M, Nv, D]``: one launch, one contiguous head-major map per layer.
    """
    B, Nv, E = value_proj_out.shape
    E = E // num_groups
    if not value_proj_out.is_cuda:
        raise RuntimeError("value_to_head_major: value must be a HIP (cuda) tensor; no CPU fallback")
    if value_proj_out.stride(2) != 1 or value_proj_out.stride(0) != Nv * value_proj_out.stride(1):
        value_proj_out = value_proj_out.contiguous()
    D = E // num_heads
    out_dtype = out_dtype or value_proj_out.dtype
    shape = (B, num_heads, Nv, D) if num_groups == 1 else (num_groups, B, num_heads, Nv, D)
    dst = torch.empty(shape, dtype=out_dtype, device=value_proj_out.device)
    mask_u8 = None
    if key_padding_mask is not None:
        _hip.require_device("value_to_head_major", key_padding_mask=key_padding_mask)
        mask_u8 = key_padding_mask.view(torch.uint8) if key_padding_mask.dtype == torch.bool else key_padding_mask
    with torch.cuda.device(dst.device):
        code = _hip.lib(value_proj_out.dtype).sdetr_value_to_head_major(
            _hip.stream_ptr(), value_proj_out.data_ptr(), _hip.dtype_code(value_proj_out.dtype),
            value_proj_out.stride(1), _hip.ptr(mask_u8), B, Nv, num_heads, D, num_groups, dst.data_ptr(),
            _hip.dtype_code(out_dtype))
    _hip.check(code, "value_to_head_major")
    return dst


def msda_fused_forward(value_hm: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                       reference_points: Tensor, proj: Tensor, num_levels: int, num_points: int,
                       order: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                       proj_head_major: bool = False) -> Tensor:
    """softmax + sampling locations + gather-reduce in one launch (ms_deform_attn.py:322-372).

    ``proj`` is the concatenated ``[sampling_offsets | attention_weights]`` projection of the query,
    ``[B, Nq, >= 3*M*L*P]``; ``reference_points`` is ``[B, Nq, L, 2|4]`` fp32.  With ``proj_head_major`` it is
    ``[B, M, Nq, 3*L*P]`` instead (per head: offsets, then logits).
    """
    _hip.require_device("msda_fused_forward", value_hm=value_hm, spatial_shapes=spatial_shapes,
                        level_start_index=level_start_index, order=order)
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))
    B, M, Nv, D = value_hm.shape
    if proj_head_major:
        if proj.dim() != 4 or proj.shape[1] != M or proj.shape[3] != 3 * num_levels * num_points or not proj.is_contiguous():
            raise RuntimeError("msda_fused_forward: head-major proj must be a contiguous [B, M, Nq, 3*L*P] tensor")
        Nq, proj_stride = proj.shape[2], 0
    else:
        Nq = proj.shape[1]
        if proj.stride(2) != 1 or proj.stride(0) != Nq * proj.stride(1):
            proj = proj.contiguous()
        proj_stride = proj.stride(1)
    if not reference_points.is_cuda:
        raise RuntimeError("msda_fused_forward: reference_points must be a HIP (cuda) tensor; no CPU fallback")
    if reference_points.dtype != torch.float32:
        reference_points = reference_points.float()
    # a row prefix [B, :Nq] of a longer reference-point buffer is accepted as is (images a batch stride apart)
    if Nq > 0 and not reference_points[0].is_contiguous():
        reference_points = reference_points.contiguous()
    ref_bs = reference_points.stride(0) if (B > 1 and Nq > 0) else 0
    out_dtype = out_dtype or proj.dtype
    out = torch.empty((B, Nq, M * D), dtype=out_dtype, device=value_hm.device)
    with torch.cuda.device(out.device):
        code = _hip.lib(proj.dtype).sdetr_msda_fused_forward(
            _hip.stream_ptr(), value_hm.data_ptr(), _hip.dtype_code(value_hm.dtype), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), reference_points.data_ptr(), reference_points.shape[-1], ref_bs,
            proj.data_ptr(), _hip.dtype_code(proj.dtype), proj_stride, 1 if proj_head_major else 0, _hip.ptr(order),
            B, Nv, M, D, num_levels, Nq, num_points, out.data_ptr(), _hip.dtype_code(out_dtype))
    _hip.check(code, "msda_fused_forward")
    return out


_RESIDENT_MAX = None


def resident_supported(value_hm: Tensor, level_shapes, num_levels: int, num_points: int) -> bool:
    """Whether ``msda_resident_forward`` (levels 2+3 of the pyramid resident in LDS, or level 3 alone when the two do not
    fit together -- the reference's 5scale pyramid) covers this call: fp16 head-major maps with 32-channel heads, 4 levels
    x 4 points, a host copy of the level shapes, and a coarsest level that fits."""
    global _RESIDENT_MAX
    if (level_shapes is None or len(level_shapes) != 4 or num_levels != 4 or num_points != 4
            or value_hm.dtype != torch.float16 or value_hm.shape[-1] != 32):
        return False
    if sum(int(h) * int(w) for h, w in level_shapes) != value_hm.shape[2]:
        return False
    if _RESIDENT_MAX is None:
        _RESIDENT_MAX = int(_hip.lib().sdetr_msda_resident_max_pixels())
    return int(level_shapes[3][0]) * int(level_shapes[3][1]) <= _RESIDENT_MAX


def msda_resident_forward(value_hm: Tensor, level_shapes, reference_points: Tensor, proj_hm: Tensor,
                          out_dtype: Optional[torch.dtype] = None, chunks: int = 0,
                          image_lanes: Optional[int] = None) -> Tensor:
    """``msda_fused_forward`` with the coarse levels served from LDS (csrc/msda_resident.hip): ``value_hm``
    ``[B,M,Nv,32]`` fp16, ``level_shapes`` HOST list of the four (h, w), ``reference_points`` ``[B,Nq,4,2|4]`` fp32,
    ``proj_hm`` the head-major bf16 projection slab ``[B,M,Nq,48]``.  ``chunks``: workgroups per (image, head),
    0 = one workgroup per CU."""
    import ctypes
    _hip.require_device("msda_resident_forward", value_hm=value_hm, proj_hm=proj_hm)
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))
    B, M, Nv, D = value_hm.shape
    if not resident_supported(value_hm, level_shapes, 4, 4):
        raise RuntimeError("msda_resident_forward: fp16 [B,M,Nv,32] maps of a 4-level pyramid whose two coarse levels "
                           "fit in LDS expected")
    if (proj_hm.dim() != 4 or proj_hm.shape[1] != M or proj_hm.shape[3] != 48 or not _hip.is_act16(proj_hm.dtype)
            or not proj_hm.is_contiguous()):
        raise RuntimeError("msda_resident_forward: proj must be a contiguous bf16 | fp16 [B, M, Nq, 48] tensor")
    if not value_hm.is_contiguous():   # the kernel addresses dense [B,M,Nv,32] maps
        raise RuntimeError("msda_resident_forward: value_hm must be contiguous [B, M, Nv, 32]")
    Nq = proj_hm.shape[2]
    if not reference_points.is_cuda:
        raise RuntimeError("msda_resident_forward: reference_points must be a HIP (cuda) tensor; no CPU fallback")
    if reference_points.dtype != torch.float32:
        reference_points = reference_points.float()
    if Nq > 0 and not reference_points[0].is_contiguous():
        reference_points = reference_points.contiguous()
    ref_bs = reference_points.stride(0) if (B > 1 and Nq > 0) else 0
    out_dtype = out_dtype or proj_hm.dtype
    out = torch.empty((B, Nq, M * D), dtype=out_dtype, device=value_hm.device)
    hw = (ctypes.c_int32 * 8)(*[int(v) for s in level_shapes for v in s])
    with torch.cuda.device(out.device):
        code = _hip.lib(proj_hm.dtype).sdetr_msda_resident_forward_ex(
            _hip.stream_ptr(), value_hm.data_ptr(), _hip.dtype_code(value_hm.dtype), hw, reference_points.data_ptr(),
            reference_points.shape[-1], ref_bs, proj_hm.data_ptr(), B, Nv, M, Nq, out.data_ptr(),
            _hip.dtype_code(out_dtype), int(chunks), -1 if image_lanes is None else int(image_lanes))
    _hip.check(code, "msda_resident_forward")
    return out


# ---- bordered head-major maps (round 4; include/salience_hip.h, csrc/msda_resident.hip msda_bordered_kernel) -------------

class BorderedLayout:
    """Geometry of the bordered head-major layout of one pyramid: level l is (H_l + 2) rows of (W_l + 1) records (row -1
    and row H_l zero, record -1 of every row zero = record W_l of the row above), the levels one after the other, one
    closing zero record.  ``records``: records per (image, head); ``pixel_map`` int32 [Nv]: record of every token (level-
    major raster order, the reference's flatten order, base_transformer.py:22-33); ``border`` int32 [nb]: the zero
    records.  Device copies are cached per device."""

    def __init__(self, level_shapes):
        self.level_shapes = [(int(h), int(w)) for h, w in level_shapes]
        starts, pix, border, p = [], [], [], 0
        for h, w in self.level_shapes:
            starts.append(p)
            rows = torch.arange(h, dtype=torch.int64).view(h, 1)
            cols = torch.arange(w, dtype=torch.int64).view(1, w)
            pix.append((p + (rows + 1) * (w + 1) + cols + 1).reshape(-1))
            border.append(p + torch.arange(w + 1, dtype=torch.int64))                       # row -1
            border.append(p + (h + 1) * (w + 1) + torch.arange(w + 1, dtype=torch.int64))   # row H
            border.append(p + (torch.arange(h, dtype=torch.int64) + 1) * (w + 1))           # record -1 of rows 0..H-1
            p += (h + 2) * (w + 1)
        border.append(torch.tensor([p], dtype=torch.int64))                                  # the closing record
        self.starts = starts
        self.records = p + 1
        self.tokens = sum(h * w for h, w in self.level_shapes)
        self.pixel_map_host = torch.cat(pix).to(torch.int32)
        self.border_host = torch.cat(border).to(torch.int32)
        self._dev = {}

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = (self.pixel_map_host.to(device), self.border_host.to(device))
        return self._dev[key]


_BORDERED_LAYOUTS = {}


def bordered_layout(level_shapes) -> BorderedLayout:
    key = tuple((int(h), int(w)) for h, w in level_shapes)
    if key not in _BORDERED_LAYOUTS:
        _BORDERED_LAYOUTS[key] = BorderedLayout(key)
    return _BORDERED_LAYOUTS[key]


_BORDERED_MAX = None


def bordered_supported(level_shapes, num_levels: int, num_points: int, channels: int = 32) -> bool:
    """Whether ``msda_bordered_forward`` covers this pyramid: 4 levels x 4 points, 32-channel heads, and a coarsest level
    whose bordered form fits the LDS next to the weight tables."""
    global _BORDERED_MAX
    if level_shapes is None or len(level_shapes) != 4 or num_levels != 4 or num_points != 4 or channels != 32:
        return False
    if _BORDERED_MAX is None:
        _BORDERED_MAX = int(_hip.lib().sdetr_msda_bordered_max_resident_records())
    h, w = level_shapes[3]
    return (int(h) + 2) * (int(w) + 1) + 1 <= _BORDERED_MAX


def is_bordered(value_hm: Tensor, level_shapes) -> bool:
    """``value_hm`` ``[..., Np, 32]`` holds the bordered layout of ``level_shapes`` (decided by its record count)."""
    if level_shapes is None or len(level_shapes) != 4:
        return False
    lay = bordered_layout(level_shapes)
    return value_hm.shape[-2] == lay.records and lay.records != lay.tokens


def to_bordered(value_hm: Tensor, level_shapes) -> Tensor:
    """Plain head-major maps ``[..., Nv, D]`` -> bordered ``[..., Np, D]`` (tests and micro-benchmarks; the product path
    writes the bordered form straight from the value projection, ``filter_ops.plan_value_projection``)."""
    lay = bordered_layout(level_shapes)
    if value_hm.shape[-2] != lay.tokens:
        raise RuntimeError("to_bordered: the maps do not hold this pyramid's tokens")
    pix, _ = lay.on(value_hm.device)
    out = torch.zeros(value_hm.shape[:-2] + (lay.records, value_hm.shape[-1]), dtype=value_hm.dtype, device=value_hm.device)
    out.index_copy_(value_hm.dim() - 2, pix.long(), value_hm)
    return out


def spatial_row_order(token_index: Tensor, level_shapes, tile: int = 16) -> Tensor:
    """A row order for ``msda_bordered_forward``: the rows (tokens ``token_index`` [B, Nq], level-major raster indices)
    sorted tile-major -- tiles of ``tile`` x ``tile`` finest-level pixels, the tokens of all levels whose centre falls
    into a tile next to each other, raster order inside.  int32 [B, Nq].  (torch reference of the kernel that builds the
    per-layer orders in the step, ``filter_ops.layer_row_orders``.)"""
    pos = tile_major_positions(level_shapes, tile).to(token_index.device)
    return pos[token_index.long()].argsort(dim=1, stable=True).to(torch.int32)


_TILE_POS = {}


def tile_major_positions(level_shapes, tile: int = 16) -> Tensor:
    """int32 [Nv]: position of every token in the tile-major order of the pyramid (host tensor, cached)."""
    key = (tuple((int(h), int(w)) for h, w in level_shapes), int(tile))
    if key not in _TILE_POS:
        h0, w0 = key[0][0]
        ty, tx = max(1, -(-h0 // tile)), max(1, -(-w0 // tile))
        keys = []
        for lvl, (h, w) in enumerate(key[0]):
            cy = ((torch.arange(h, dtype=torch.float64) + 0.5) / h * ty).floor().clamp_(max=ty - 1).long()
            cx = ((torch.arange(w, dtype=torch.float64) + 0.5) / w * tx).floor().clamp_(max=tx - 1).long()
            t = cy.view(h, 1) * tx + cx.view(1, w)                                    # tile of the token's centre
            # inside a tile: level, then raster
            inner = lvl * (1 << 20) + torch.arange(h * w, dtype=torch.int64).view(h, w)
            keys.append((t * (1 << 24) + inner).reshape(-1))
        k = torch.cat(keys)
        order = k.argsort(stable=True)
        pos = torch.empty_like(order)
        pos[order] = torch.arange(order.numel(), dtype=torch.int64)
        _TILE_POS[key] = pos.to(torch.int32)
    return _TILE_POS[key]


ACC_DEFAULT, ACC_EXACT, ACC_PACKED_SAMPLE, ACC_PACKED_LEVEL = -1, 0, 1, 2      # SDETR_MSDA_ACC_* (include/salience_hip.h)


def msda_bordered_forward(value_bordered: Tensor, level_shapes, reference_points: Tensor, proj_hm: Tensor,
                          row_order: Optional[Tensor] = None, out_dtype: Optional[torch.dtype] = None,
                          chunks: int = 0, accumulate: int = ACC_DEFAULT, image_lanes: Optional[int] = None,
                          l2_warmup: Optional[int] = None) -> Tensor:
    """``msda_resident_forward`` on bordered maps ``[B,M,Np,32]`` fp16 (``to_bordered`` / the value projection's bordered
    store); ``row_order`` optional int32 ``[B,Nq]`` permutation of the rows (processing order only).  ``accumulate``
    (``ACC_*``): how a 16-bit output's corner products are summed -- ``ACC_EXACT`` = fp32 as the reference's op,
    ``ACC_DEFAULT`` = the library's choice for the activation type; ``image_lanes`` / ``l2_warmup``: launch choices
    (``None`` = the library's rule), see ``sdetr_msda_bordered_forward_ex``."""
    import ctypes
    _hip.require_device("msda_bordered_forward", value_bordered=value_bordered, proj_hm=proj_hm)
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
            reference_points.shape[-1]))
    B, M, Np, D = value_bordered.shape
    if (not bordered_supported(level_shapes, 4, 4, D) or value_bordered.dtype != torch.float16
            or not is_bordered(value_bordered, level_shapes)):
        raise RuntimeError("msda_bordered_forward: fp16 bordered [B,M,Np,32] maps of a 4-level pyramid whose coarsest level "
                           "fits in LDS expected")
    if (proj_hm.dim() != 4 or proj_hm.shape[1] != M or proj_hm.shape[3] != 48 or not _hip.is_act16(proj_hm.dtype)
            or proj_hm.shape[0] != B):
        raise RuntimeError("msda_bordered_forward: proj must be a contiguous bf16 | fp16 [B, M, Nq, 48] tensor")
    Nq = proj_hm.shape[2]
    if row_order is not None and (not row_order.is_cuda or row_order.dtype != torch.int32 or row_order.shape != (B, Nq)
                                  or (Nq > 1 and row_order.stride(1) != 1)):
        raise RuntimeError("msda_bordered_forward: row_order must be a device int32 [B, Nq] with contiguous rows")
    if not reference_points.is_cuda:
        raise RuntimeError("msda_bordered_forward: reference_points must be a HIP (cuda) tensor; no CPU fallback")
    if reference_points.dtype != torch.float32:
        reference_points = reference_points.float()
    if Nq > 0 and not reference_points[0].is_contiguous():
        reference_points = reference_points.contiguous()
    ref_bs = reference_points.stride(0) if (B > 1 and Nq > 0) else 0
    out_dtype = out_dtype or proj_hm.dtype
    out = torch.empty((B, Nq, M * D), dtype=out_dtype, device=value_bordered.device)
    hw = (ctypes.c_int32 * 8)(*[int(v) for s in level_shapes for v in s])
    with torch.cuda.device(out.device):
        code = _hip.lib(proj_hm.dtype).sdetr_msda_bordered_forward_ex(
            _hip.stream_ptr(), value_bordered.data_ptr(), _hip.dtype_code(value_bordered.dtype), hw,
            reference_points.data_ptr(), reference_points.shape[-1], ref_bs, proj_hm.data_ptr(), _hip.ptr(row_order),
            (row_order.stride(0) if B > 1 else Nq) if row_order is not None else 0, B, Np, M, Nq, out.data_ptr(),
            _hip.dtype_code(out_dtype), int(chunks), int(accumulate), -1 if image_lanes is None else int(image_lanes),
            -1 if l2_warmup is None else int(l2_warmup))
    _hip.check(code, "msda_bordered_forward")
    return out


def last_forward_kernel(act: Optional[torch.dtype] = None) -> int:
    """``SDETR_KERNEL_*`` code of the kernel the calling thread's last MSDA forward call dispatched to (``act``:
    ``torch.float16`` asks the fp16-activation library, which keeps its own record)."""
    return int(_hip.lib(act).sdetr_msda_last_kernel())


KERNEL_GENERIC, KERNEL_GATHER, KERNEL_L4P4, KERNEL_RESIDENT, KERNEL_BORDERED, KERNEL_BORDERED_ORDERED = 1, 2, 3, 4, 5, 6

def msda_forward_head_major(value_hm: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                            sampling_loc: Tensor, attn_weight: Tensor,
                            out_dtype: torch.dtype = torch.float32) -> Tensor:
    """The reference op's math on the head-major value layout (explicit fp32 locations / weights)."""
    _hip.require_device("msda_forward_head_major", value_hm=value_hm, spatial_shapes=spatial_shapes,
                        level_start_index=level_start_index, sampling_loc=sampling_loc, attn_weight=attn_weight)
    B, M, Nv, D = value_hm.shape
    _, Nq, _, L, P, _ = sampling_loc.shape
    out = torch.empty((B, Nq, M * D), dtype=out_dtype, device=value_hm.device)
    with torch.cuda.device(out.device):
        code = _hip.lib(out_dtype).sdetr_msda_forward_head_major(
            _hip.stream_ptr(), value_hm.data_ptr(), _hip.dtype_code(value_hm.dtype), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            B, Nv, M, D, L, Nq, P, out.data_ptr(), _hip.dtype_code(out_dtype))
    _hip.check(code, "msda_forward_head_major")
    return out


def _stacked_value_proj(attn_modules):
    """Concatenated ``value_proj`` weights / biases of the modules, cached on the first one."""
    first = attn_modules[0]
    owner = first.__dict__
    ps = [p for m in attn_modules for p in (m.value_proj.weight, m.value_proj.bias)]
    key = tuple((p.data_ptr(), p._version, p.dtype) for p in ps)
    hit = owner.get("_batched_value_proj")
    if hit is None or hit[0] != key:
        w = torch.cat([m.value_proj.weight.detach() for m in attn_modules], 0).contiguous()
        b = torch.cat([m.value_proj.bias.detach() for m in attn_modules], 0).contiguous()
        hit = (key, w, b)
        owner["_batched_value_proj"] = hit
    return hit[1], hit[2]


def _bordered_levels_for(attn_modules, level_shapes, vdt):
    """The level shapes if these modules' maps can take the bordered layout (fp16 maps, 4 levels x 4 points, 32-channel
    heads, a coarsest level that fits the LDS), else None."""
    first = attn_modules[0]
    if (level_shapes is None or vdt != torch.float16 or first.embed_dim != 32 * first.num_heads
            or not all(m.num_levels == 4 and m.num_points == 4 for m in attn_modules)
            or not bordered_supported(level_shapes, 4, 4)):
        return None
    return [(int(h), int(w)) for h, w in level_shapes]


def plan_batched_value_maps(attn_modules, value: Tensor, padding_mask: Optional[Tensor], parts=2, level_shapes=None):
    """``batched_value_maps`` as pending jobs: ``(maps [n,B,M,Nv,D], [ValueProjectionJob, ...])`` -- slices of the one
    projection that other launches can carry (``filter_ops.salience_head(value_job=...)``) -- or ``None`` when the
    one-launch kernel does not cover the configuration (call ``batched_value_maps`` then)."""
    from .filter_ops import plan_value_projection, token_linear_applies
    first = attn_modules[0]
    heads, E = first.num_heads, first.embed_dim
    w_all, b_all = _stacked_value_proj(attn_modules)
    vdt = first.value_dtype or value.dtype
    if not (token_linear_applies(value, w_all) and E == 32 * heads and vdt in (torch.float16, torch.bfloat16)
            and value.is_contiguous() and value.dim() == 3):
        return None
    return plan_value_projection(value, w_all, b_all, padding_mask, heads, len(attn_modules), vdt, parts=parts,
                                 bordered_levels=_bordered_levels_for(attn_modules, level_shapes, vdt))


def batched_value_maps(attn_modules, value: Tensor, padding_mask: Optional[Tensor], level_shapes=None) -> Tensor:
    """Head-major value maps ``[len(attn_modules), B, M, Nv, D]`` of several ``MultiScaleDeformableAttention``
    modules that sample the SAME ``value`` (the six encoder layers, salience_transformer.py:452; the decoder layers'
    cross-attentions, :575-582): their ``value_proj`` run as one projection.  No-grad path only."""
    from .filter_ops import token_linear_applies, value_proj_head_major
    first = attn_modules[0]
    heads, E = first.num_heads, first.embed_dim
    w_all, b_all = _stacked_value_proj(attn_modules)
    vdt = first.value_dtype or value.dtype
    n = len(attn_modules)
    if (token_linear_applies(value, w_all) and E == 32 * heads and vdt in (torch.float16, torch.bfloat16)
            and value.is_contiguous()):
        # projection, padding mask, 16-bit conversion and head-major layout in one launch
        # (with the pyramid's level shapes: the bordered layout the round-4 MSDA kernel reads, [n,B,M,Np,D])
        return value_proj_head_major(value, w_all, b_all, padding_mask, heads, n, vdt,
                                     bordered_levels=_bordered_levels_for(attn_modules, level_shapes, vdt))
    v_all = F.linear(value, w_all, b_all)                      # [B, Nv, n*E]
    out = value_to_head_major(v_all, padding_mask, heads, vdt, num_groups=n)
    return out[None] if n == 1 else out


def invalidate_caches(module: nn.Module) -> None:
    """Drop every derived-weight cache below ``module`` (fused / head-major projection operands, batched value
    projections, packed MFMA operands of the Linear layers, folded neck plans, flattened background tables).

    The caches are keyed on ``(data_ptr, _version)`` of their parameters, which ``optimizer.step()``, ``load_state_dict``,
    ``.to()`` and every in-place op on the parameter change.  A write THROUGH ``p.data`` (``p.data.copy_()``, EMA helpers,
    ``nn.init`` on ``.data``) changes neither: call this function afterwards, or write ``with torch.no_grad(): p.copy_(..)``.
    """
    for m in module.modules():
        for attr in ("_fused_cache", "_fused_hm_cache", "_plan", "_flat_cache", "_enc_output_cast"):
            if attr in m.__dict__:
                m.__dict__[attr] = None
        m.__dict__.pop("_batched_value_proj", None)
        for p in m.parameters(recurse=False):
            for key in ("_sdetr_packed", "_sdetr_ffn", "_sdetr_tl", "_sdetr_tl512", "_sdetr_f32"):
                p.__dict__.pop(key, None)


_EXCLUSIVE = threading.local()


def _mark_exclusive(t: Tensor) -> None:
    """Record the storage of a gradient its producer has just allocated (consumed once, by _ZeroRowsInPlace.backward on
    the autograd thread that produced it)."""
    s = getattr(_EXCLUSIVE, "ptrs", None)
    if s is None:
        s = _EXCLUSIVE.ptrs = set()
    if len(s) > 64:          # (marks nobody consumed: gradients that went elsewhere)
        s.clear()
    s.add(t.untyped_storage().data_ptr())


def _take_exclusive(t: Tensor) -> bool:
    s = getattr(_EXCLUSIVE, "ptrs", None)
    p = t.untyped_storage().data_ptr()
    if s is not None and p in s:
        s.discard(p)
        return True
    return False


class _ZeroRowsInPlace(Function):
    """``x.masked_fill_(mask[..., None], 0)`` whose backward masks the incoming gradient in place as well."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        ctx.mark_dirty(x)
        return x.masked_fill_(mask[..., None], 0.0)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad):
        (mask,) = ctx.saved_tensors
        m = mask.reshape(mask.shape + (1,) * (grad.dim() - mask.dim()))   # (reshape: the mask may be a slice / transpose)
        # In place only for the one producer this Function is used behind: the value operand of
        # MultiScaleDeformableAttnFunction, whose backward hands over a grad_value it has just allocated (a [B,Nv,M,D]
        # view of it arrives here: nobody else holds it).  Anything else -- a non-contiguous gradient, a tensor some hook
        # retained (its version counter or base say so) -- is masked out of place, as autograd requires.
        # Out of place by default; in place only when the producer marked the storage as exclusively owned
        # (MultiScaleDeformableAttnFunction.backward) AND nothing has written to it since.
        fresh = (_take_exclusive(grad) and grad.is_contiguous() and not grad.requires_grad and grad._version <= 1
                 and (grad._base is None or grad._base._version <= 1))
        return (grad.masked_fill_(m, 0.0) if fresh else grad.masked_fill(m, 0.0)), None


class _SamplingPrep(Function):
    """``(sampling_locations, attention_weights)`` of ms_deform_attn.py:322-349 from the two Linear outputs and the
    reference points (``csrc/sampling_prep.hip``); the reference points carry no gradient here."""

    @staticmethod
    def applies(offsets: Tensor, logits: Tensor, reference_points: Tensor, L: int, P: int) -> bool:
        # dense operands of exactly the shapes the kernel indexes (a broadcastable [B,Nq,1,2] reference, host tensors or
        # mismatched logits take the torch formula instead -- ADVICE r3)
        if not (offsets.is_cuda and logits.is_cuda and reference_points.is_cuda and offsets.dim() == 6 and offsets.numel() > 0):
            return False
        B, Nq, M = offsets.shape[:3]
        return (offsets.dtype == torch.float32 and logits.dtype == torch.float32
                and reference_points.dtype == torch.float32 and not reference_points.requires_grad
                and tuple(offsets.shape) == (B, Nq, M, L, P, 2) and logits.numel() == B * Nq * M * L * P
                and reference_points.dim() == 4 and tuple(reference_points.shape[:3]) == (B, Nq, L)
                and reference_points.shape[3] in (2, 4)
                and bool(_hip.lib().sdetr_sampling_prep_supported(L, P)))

    @staticmethod
    def forward(ctx, offsets, logits, reference_points, spatial_shapes, L, P):
        B, Nq, M = offsets.shape[:3]
        off, lg = offsets.contiguous(), logits.contiguous()
        ref = reference_points.contiguous()
        RD = ref.shape[-1]
        loc = torch.empty((B, Nq, M, L, P, 2), dtype=torch.float32, device=off.device)
        w = torch.empty((B, Nq, M, L, P), dtype=torch.float32, device=off.device)
        with torch.cuda.device(off.device):
            code = _hip.lib().sdetr_sampling_prep_f32(_hip.stream_ptr(), off.data_ptr(), lg.data_ptr(), ref.data_ptr(),
                                                      spatial_shapes.data_ptr(), B * Nq, M, L, P, RD, loc.data_ptr(),
                                                      w.data_ptr())
        _hip.check(code, "sampling_prep")
        ctx.save_for_backward(w, ref, spatial_shapes)
        ctx.dims = (B, Nq, M, L, P, RD)
        return loc, w

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loc, grad_w):
        w, ref, spatial_shapes = ctx.saved_tensors
        B, Nq, M, L, P, RD = ctx.dims
        gl, gw = grad_loc.contiguous(), grad_w.contiguous()
        g_off = torch.empty((B, Nq, M, L, P, 2), dtype=torch.float32, device=w.device)
        g_lg = torch.empty((B, Nq, M, L * P), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            code = _hip.lib().sdetr_sampling_prep_backward_f32(
                _hip.stream_ptr(), gl.data_ptr(), gw.data_ptr(), w.data_ptr(), ref.data_ptr(), spatial_shapes.data_ptr(),
                B * Nq, M, L, P, RD, g_off.data_ptr(), g_lg.data_ptr())
        _hip.check(code, "sampling_prep_backward")
        return g_off, g_lg, None, None, None, None


class MultiScaleDeformableAttention(nn.Module):
    """Multi-Scale Deformable Attention Module (Deformable-DETR), MI355X-native inside.

    Constructor / parameters / ``forward`` signature as the reference class
    (``models/bricks/ms_deform_attn.py:215-294``).  ``value_dtype`` selects the storage type of the
    head-major value map on the no-grad path (``None`` = the activation dtype).
    """

    def __init__(self, embed_dim: int = 256, num_levels: int = 4, num_heads: int = 8, num_points: int = 4,
                 img2col_step: int = 64, value_dtype: Optional[torch.dtype] = None):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError(
                "embed_dim must be divisible by num_heads, but got {} and {}".format(embed_dim, num_heads))
        head_dim = embed_dim // num_heads
        if not _is_power_of_2(head_dim):
            warnings.warn("You'd better set embed_dim in MSDeformAttn to make sure that each dim of the "
                          "attention head a power of 2, which is more efficient.")
        self.im2col_step = img2col_step
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_levels = num_levels
        self.num_points = num_points
        self.value_dtype = value_dtype
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self._fused_cache = None
        self.init_weights()

    def init_weights(self):
        """Default initialisation (ms_deform_attn.py:262-284): zero offset weights, 8-direction ring
        bias scaled by the point index, uniform attention, xavier value/output projections."""
        # in-place on the parameters themselves (under no_grad inside nn.init): this bumps their version counters, which
        # the derived-weight caches of the native path are keyed on -- writes through `.data` would not
        constant_(self.sampling_offsets.weight, 0.0)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2)
        grid = grid.repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.view(-1))
        constant_(self.attention_weights.weight, 0.0)
        constant_(self.attention_weights.bias, 0.0)
        xavier_uniform_(self.value_proj.weight)
        constant_(self.value_proj.bias, 0.0)
        xavier_uniform_(self.output_proj.weight)
        constant_(self.output_proj.bias, 0.0)
        invalidate_caches(self)

    # -- native path pieces --------------------------------------------------------------------
    def _fused_query_projection(self):
        """[sampling_offsets ; attention_weights] as one (384 x 256) GEMM operand, cached per
        parameter version so eval-mode forwards do not re-concatenate."""
        ps = (self.sampling_offsets.weight, self.sampling_offsets.bias,
              self.attention_weights.weight, self.attention_weights.bias)
        key = tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in ps)
        if self._fused_cache is None or self._fused_cache[0] != key:
            w = torch.cat([ps[0].detach(), ps[2].detach()], 0).contiguous()
            b = torch.cat([ps[1].detach(), ps[3].detach()], 0).contiguous()
            self._fused_cache = (key, w, b)
        return self._fused_cache[1], self._fused_cache[2]

    def _fused_query_projection_head_major(self):
        """The same operand with its rows ordered by head -- head m: its 2*L*P offset rows, then its L*P logit rows --
        so that the token-resident projection kernel can store ``[B, M, Nq, 3*L*P]`` directly."""
        w, b = self._fused_query_projection()
        hit = getattr(self, "_fused_hm_cache", None)
        if hit is None or hit[0] is not w:
            M, LP = self.num_heads, self.num_levels * self.num_points
            order = torch.cat([torch.cat([torch.arange(m * 2 * LP, (m + 1) * 2 * LP),
                                          M * 2 * LP + torch.arange(m * LP, (m + 1) * LP)]) for m in range(M)]).to(w.device)
            hit = (w, w[order].contiguous(), b[order].contiguous())
            self._fused_hm_cache = hit
        return hit[1], hit[2]

    def project_value(self, value: Tensor, key_padding_mask: Optional[Tensor]) -> Tensor:
        """value_proj + padding zero-fill + head-major re-layout -> ``[B, M, Nv, D]``."""
        v = F.linear(value, self.value_proj.weight, self.value_proj.bias)
        return value_to_head_major(v, key_padding_mask, self.num_heads, self.value_dtype or v.dtype)

    # Round 5: the round-2/3 coarse-levels-in-LDS kernel on PLAIN maps (msda_resident_kernel) is retired from the module's
    # dispatch -- every caller with a host copy of the level shapes gets bordered maps and msda_bordered_kernel, plain maps
    # take the direct gather.  ``msda_resident_forward`` stays as an entry point (sdetr_msda_resident_forward, a test
    # cross-check of the bordered kernel); an integer here (queries per image) switches the old dispatch back on.
    resident_min_queries = None
    # launch choices of the bordered gather (sdetr_msda_bordered_forward_ex): ACC_DEFAULT / None = the library's rules
    bordered_accumulate = ACC_DEFAULT
    bordered_l2_warmup = None

    def head_major_projection_applies(self, query: Tensor, value_hm: Tensor) -> bool:
        """``forward_native`` (no ``order``) takes the per-head projection slabs for this input."""
        from .filter_ops import token_linear_applies
        w, _ = self._fused_query_projection()
        return (query.dim() == 3 and query.shape[0] * query.shape[1] >= 3000 and token_linear_applies(query, w)
                and self.num_levels == 4 and self.num_points == 4 and self.num_heads == 8 and value_hm.shape[-1] == 32
                and value_hm.dtype in (torch.float16, torch.bfloat16))

    def forward_native(self, query: Tensor, reference_points: Tensor, value_hm: Tensor, spatial_shapes: Tensor,
                       level_start_index: Tensor, order: Optional[Tensor] = None,
                       query_pos: Optional[Tensor] = None, apply_output_proj: bool = True,
                       level_shapes=None, head_major_projection: Optional[Tensor] = None,
                       row_order: Optional[Tensor] = None) -> Tensor:
        """``query_pos`` (optional): position embedding still to be added to ``query`` -- folded into the projection
        kernel's prologue on the bf16 path.  ``apply_output_proj=False`` returns the sampled heads ``[B,Nq,E]`` for a
        caller that fuses ``output_proj`` with what follows it.  ``head_major_projection``: the offset | weight
        projection ``[B,M,Nq,48]`` if another launch already produced it (``head_major_projection_applies`` says when
        this method would take that form)."""
        from .filter_ops import rows_linear, rows_linear_applies, token_linear, token_linear_applies
        w, b = self._fused_query_projection()
        # (the token-resident kernel's run time is flat in the token count, ~17 us; below ~12 000 tokens the library GEMM
        # behind an elementwise add is 2-3 us faster, but only the resident kernel writes the per-head slabs that save
        # the gather kernel as much -- so it is used down to a few thousand tokens)
        bordered = is_bordered(value_hm, level_shapes)   # maps in the bordered layout (the encoder's batched projection)
        if bordered and not (query.dim() == 3 and token_linear_applies(query, w) and order is None):
            raise RuntimeError("MultiScaleDeformableAttention.forward_native: bordered value maps need contiguous bf16 "
                               "[B,Nq,256] queries (the head-major projection path)")
        big = query.dim() == 3 and (bordered or query.shape[0] * query.shape[1] >= 3000)
        head_major = (big and token_linear_applies(query, w) and order is None and self.num_levels == 4
                      and self.num_points == 4 and value_hm.shape[-1] == 32
                      and value_hm.dtype in (torch.float16, torch.bfloat16)
                     )
        if head_major:
            # per-head slabs: every XCD's L2 then fetches only its own head's projection values
            proj = head_major_projection
            if proj is None:
                wh, bh = self._fused_query_projection_head_major()
                proj = token_linear(query, wh, bh, x_add=query_pos, group_features=3 * self.num_levels * self.num_points)
            if bordered:
                # round 4: zero-bordered maps + (optional) spatial row order -- every layer size takes this kernel
                out = msda_bordered_forward(value_hm, level_shapes, reference_points, proj, row_order=row_order,
                                            out_dtype=query.dtype, accumulate=self.bordered_accumulate,
                                            l2_warmup=self.bordered_l2_warmup)
            elif (self.resident_min_queries is not None and query.shape[1] >= self.resident_min_queries
                    and resident_supported(value_hm, level_shapes, self.num_levels, self.num_points)):
                out = msda_resident_forward(value_hm, level_shapes, reference_points, proj, out_dtype=query.dtype)
            else:
                out = msda_fused_forward(value_hm, spatial_shapes, level_start_index, reference_points, proj,
                                         self.num_levels, self.num_points, out_dtype=query.dtype, proj_head_major=True)
            if not apply_output_proj:
                return out
            return F.linear(out, self.output_proj.weight, self.output_proj.bias)
        if big and token_linear_applies(query, w):
            proj = token_linear(query, w, b, x_add=query_pos)
        elif query_pos is not None and rows_linear_applies(query, w, b) and query_pos.shape == query.shape:
            # a few thousand rows (the decoder's 900 queries per image): add + projection in one launch
            # (every output sees x + pos; the kernel selects per 32-feature tile and accepts a count rounded up to the
            # tile, so that output widths that are not multiples of 32 -- levels * points % 4 != 0 -- take it too: ADVICE r5)
            proj = rows_linear(query, w, b, pos=query_pos, pos_features=(w.shape[0] + 31) // 32 * 32)
        else:
            proj = F.linear(query if query_pos is None else query + query_pos, w, b)
        out = msda_fused_forward(value_hm, spatial_shapes, level_start_index, reference_points, proj,
                                 self.num_levels, self.num_points, order=order, out_dtype=query.dtype)
        if not apply_output_proj:
            return out
        return F.linear(out, self.output_proj.weight, self.output_proj.bias)

    # -- reference signature -------------------------------------------------------------------
    def forward(self, query: Tensor, reference_points: Tensor, value: Tensor, spatial_shapes: Tensor,
                level_start_index: Tensor, key_padding_mask: Tensor) -> Tensor:
        """Same contract as the reference ``forward`` (ms_deform_attn.py:286-377):
        query ``[B,Nq,E]``, reference_points ``[B,Nq,L,2|4]`` in [0,1], value ``[B,Nv,E]``,
        spatial_shapes ``[L,2]`` (h,w), level_start_index ``[L]``, key_padding_mask ``[B,Nv]`` or None."""
        batch_size, num_query, _ = query.shape
        batch_size, num_value, _ = value.shape
        if not query.is_cuda:
            raise RuntimeError("MultiScaleDeformableAttention: the HIP extension path needs device tensors; "
                               "there is no CPU fallback (the CPU restatement lives in oracle/)")
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        needs_grad = torch.is_grad_enabled() and (
            query.requires_grad or value.requires_grad or reference_points.requires_grad
            or any(p.requires_grad for p in self.parameters()))
        if not needs_grad:
            value_hm = self.project_value(value, key_padding_mask)
            return self.forward_native(query, reference_points, value_hm, spatial_shapes, level_start_index)

        # autograd path: reference layout, fp32 op with the HIP forward/backward kernels
        value = self.value_proj(value)
        if key_padding_mask is not None:
            # in place, both ways: the projection's output is a fresh tensor nothing else holds and no backward needs, and
            # the gradient that comes back is the op's own fresh grad_value (the framework's masked_fill backward copies
            # the [B, Nv, 256] gradient before it masks it)
            value = _ZeroRowsInPlace.apply(value, key_padding_mask)
        value = value.view(batch_size, num_value, self.num_heads, self.embed_dim // self.num_heads)
        sampling_offsets = self.sampling_offsets(query).view(
            batch_size, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(
            batch_size, num_query, self.num_heads, self.num_levels * self.num_points)
        if _SamplingPrep.applies(sampling_offsets, attention_weights, reference_points, self.num_levels, self.num_points):
            # softmax, offset normalisation and the reference-point add in one launch (and one launch backward)
            sampling_locations, attention_weights = _SamplingPrep.apply(
                sampling_offsets, attention_weights, reference_points, spatial_shapes, self.num_levels, self.num_points)
            output = MultiScaleDeformableAttnFunction.apply(
                value.to(torch.float32).contiguous(), spatial_shapes, level_start_index, sampling_locations,
                attention_weights, self.im2col_step)
            return self.output_proj(output)
        attention_weights = attention_weights.softmax(-1).view(
            batch_size, num_query, self.num_heads, self.num_levels, self.num_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            sampling_locations = (reference_points[:, :, None, :, None, :]
                                  + sampling_offsets / offset_normalizer[None, None, None, :, None, :])
        else:
            sampling_locations = (reference_points[:, :, None, :, None, :2]
                                  + sampling_offsets / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5)
        output = MultiScaleDeformableAttnFunction.apply(
            value.to(torch.float32).contiguous(), spatial_shapes, level_start_index,
            sampling_locations.to(torch.float32).contiguous(), attention_weights.to(torch.float32).contiguous(),
            self.im2col_step)
        if value.dtype != torch.float32:
            output = output.to(value.dtype)
        return self.output_proj(output)
