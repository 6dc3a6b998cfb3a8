"""Host wrappers of the filtering / row-movement kernels (C ABI sections (4) and (5) of
include/salience_hip.h).  PyTorch only owns the memory and the stream."""
from typing import Optional

import torch
from torch import Tensor

from . import _hip


def masked_topk_desc(score: Tensor, k: int, mask: Optional[Tensor] = None, fill_with_global_min: bool = False,
                     payload: Optional[Tensor] = None, index_offset: int = 0, want_scores: bool = True):
    """Sorted-descending top-k per row with ties -> lower index first.

    ``score`` [B,N] fp32.  With ``mask`` (bool [B,N], True = masked) and ``fill_with_global_min`` the
    masked entries compete with the value ``score.min()`` taken over the WHOLE array, as the reference
    does with ``masked_fill(mask, score.min())`` before ``topk`` (salience_transformer.py:146-150).
    Returns ``(values [B,k] or None, indices [B,k] int64)``; ``indices`` are ``payload[b, pos]`` when a
    payload is given (the index gather after the global sort, :156-158), else ``pos + index_offset``.
    """
    _hip.require_device("masked_topk_desc", score=score, mask=mask, payload=payload)
    if score.dtype != torch.float32 or score.dim() != 2:
        raise RuntimeError("masked_topk_desc: score must be a 2-d float32 tensor")
    B, N = score.shape
    k = int(k)
    if k < 0 or k > N:
        # torch.topk: "selected index k out of range"
        raise RuntimeError(f"masked_topk_desc: selected index k out of range (k={k}, row length {N})")
    if mask is not None and not fill_with_global_min:
        raise RuntimeError("masked_topk_desc: a mask requires fill_with_global_min=True")
    if mask is not None:
        if mask.shape != score.shape:
            raise RuntimeError("masked_topk_desc: mask shape mismatch")
        mask = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    if payload is not None and (payload.dtype != torch.int64 or payload.shape != score.shape):
        raise RuntimeError("masked_topk_desc: payload must be int64 with score's shape")
    out_score = torch.empty((B, k), dtype=torch.float32, device=score.device) if want_scores else None
    out_index = torch.empty((B, k), dtype=torch.int64, device=score.device)
    lib = _hip.lib()
    ws_bytes = lib.sdetr_topk_workspace_bytes(B, N, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=score.device) if ws_bytes else None
    with torch.cuda.device(score.device):
        code = lib.sdetr_masked_topk_desc_f32(
            _hip.stream_ptr(), score.data_ptr(), _hip.ptr(mask), 1 if fill_with_global_min else 0,
            _hip.ptr(payload), B, N, k, int(index_offset), _hip.ptr(out_score), out_index.data_ptr(),
            _hip.ptr(ws), ws_bytes)
    _hip.check(code, "masked_topk_desc")
    return out_score, out_index


def gather_rows(src: Tensor, idx: Tensor) -> Tensor:
    """``dst[b, i] = src[b, idx[b, i]]`` for ``src`` [B,S,C], ``idx`` [B,n] int64
    (the torch.gather calls of salience_transformer.py:454-461 without the expanded index)."""
    _hip.require_device("gather_rows", src=src, idx=idx)
    if idx.dtype != torch.int64 or src.dim() < 2 or idx.dim() != 2 or idx.shape[0] != src.shape[0]:
        raise RuntimeError("gather_rows: src [B,S,...], idx [B,n] int64 expected")
    B, S = src.shape[:2]
    n = idx.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size() if S > 0 else 0
    dst = torch.empty((B, n) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    if B * n == 0:
        return dst
    with torch.cuda.device(src.device):
        code = _hip.lib().sdetr_gather_rows(_hip.stream_ptr(), src.data_ptr(), idx.data_ptr(), B, S, n, row_bytes,
                                            dst.data_ptr())
    _hip.check(code, "gather_rows")
    return dst


def scatter_rows_(dst: Tensor, idx: Tensor, src: Tensor, count: Optional[Tensor] = None) -> Tensor:
    """In place ``dst[b, idx[b, i]] = src[b, i]`` for ``i < count[b]`` (all rows when ``count`` is None):
    the per-image scatter of salience_transformer.py:474-485 with ``focus_token_nums`` read on-device."""
    _hip.require_device("scatter_rows_", dst=dst, idx=idx, src=src, count=count)
    if idx.dtype != torch.int64 or dst.dtype != src.dtype or idx.shape[0] != dst.shape[0]:
        raise RuntimeError("scatter_rows_: dst [B,S,...], idx [B,n] int64, src [B,n,...] expected")
    if count is not None and (count.dtype != torch.int64 or count.numel() != dst.shape[0]):
        raise RuntimeError("scatter_rows_: count must be int64 [B]")
    B, S = dst.shape[:2]
    n = idx.shape[1]
    if B * n == 0:
        return dst
    row_bytes = dst[0, 0].numel() * dst.element_size()
    with torch.cuda.device(dst.device):
        code = _hip.lib().sdetr_scatter_rows(_hip.stream_ptr(), dst.data_ptr(), idx.data_ptr(), src.data_ptr(),
                                             _hip.ptr(count), B, S, n, row_bytes)
    _hip.check(code, "scatter_rows_")
    return dst


def pyramid_flatten(multi_level_feats, multi_level_pos_embeds, multi_level_masks, level_embeds: Tensor,
                    want_bf16: bool = False):
    """F0 in one launch per level: ``flatten_multi_level`` + ``get_lvl_pos_embed``
    (base_transformer.py:22-33) + the token validity of ``gen_encoder_output_proposals`` (:74-112).

    Returns ``(feat_flatten [B,S,C], lvl_pos_embed_flatten [B,S,C], enc_output_input [B,S,C] =
    (feat + pos) * keep, mask_flatten [B,S] bool, feat_bf16 | None, pos_bf16 | None, valid_ratios [B,L,2])``;
    ``valid_ratios`` is ``get_valid_ratios`` of every level (base_transformer.py:48-56), a by-product of the
    extents the kernel counts anyway.
    """
    feats = [f.contiguous() for f in multi_level_feats]
    pos = [p.contiguous() for p in multi_level_pos_embeds]
    masks = [m.contiguous() for m in multi_level_masks]
    _hip.require_device("pyramid_flatten", level_embeds=level_embeds, **{f"feat{i}": f for i, f in enumerate(feats)})
    if feats[0].dtype != torch.float32 or pos[0].dtype != torch.float32:
        raise RuntimeError("pyramid_flatten: float32 feature / position maps expected")
    B, C = feats[0].shape[:2]
    S = sum(int(f.shape[2]) * int(f.shape[3]) for f in feats)
    dev = feats[0].device
    feat_out = torch.empty((B, S, C), dtype=torch.float32, device=dev)
    pos_out = torch.empty_like(feat_out)
    sum_out = torch.empty_like(feat_out)
    mask_out = torch.empty((B, S), dtype=torch.bool, device=dev)
    feat_bf16 = torch.empty((B, S, C), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    pos_bf16 = torch.empty((B, S, C), dtype=torch.bfloat16, device=dev) if want_bf16 else None
    le = level_embeds.detach().float().contiguous()
    valid_ratios = torch.empty((B, len(feats), 2), dtype=torch.float32, device=dev)
    lib = _hip.lib()
    start = 0
    with torch.cuda.device(dev):
        for lvl, (f, p, m) in enumerate(zip(feats, pos, masks)):
            H, W = int(f.shape[2]), int(f.shape[3])
            mu8 = m.view(torch.uint8) if m.dtype == torch.bool else m
            code = lib.sdetr_pyramid_flatten_level(
                _hip.stream_ptr(), f.data_ptr(), p.data_ptr(), mu8.data_ptr(), le[lvl].data_ptr(), B, C, H, W, lvl,
                start, S, feat_out.data_ptr(), pos_out.data_ptr(), sum_out.data_ptr(), mask_out.data_ptr(),
                _hip.ptr(feat_bf16), _hip.ptr(pos_bf16), valid_ratios.data_ptr() + lvl * 8, len(feats) * 2)
            _hip.check(code, "pyramid_flatten_level")
            start += H * W
    return feat_out, pos_out, sum_out, mask_out, feat_bf16, pos_bf16, valid_ratios


def class_max_times(score: Tensor, scale: Tensor) -> Tensor:
    """``score.max(-1)[0] * scale`` in one pass: score ``[B,Nq,num_classes]`` (fp32 | bf16), scale ``[B,Nq]``
    fp32 -> fp32 ``[B,Nq]`` (mc_score of salience_transformer.py:366)."""
    _hip.require_device("class_max_times", score=score, scale=scale)
    if scale.dtype != torch.float32:
        scale = scale.float()
    B, Nq, C = score.shape
    out = torch.empty((B, Nq), dtype=torch.float32, device=score.device)
    with torch.cuda.device(score.device):
        code = _hip.lib().sdetr_class_max_times(_hip.stream_ptr(), score.data_ptr(), _hip.dtype_code(score.dtype),
                                                scale.data_ptr(), B * Nq, C, out.data_ptr())
    _hip.check(code, "class_max_times")
    return out


def _bn_view(t: Tensor):
    """(tensor viewed as [B, n, C] with a contiguous last dim, batch stride, row stride)."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        t = t.reshape(-1, t.shape[-2], t.shape[-1])
    if t.stride(2) != 1:
        t = t.contiguous()
    return t, t.stride(0), t.stride(1)


def fused_layer_norm(x: Tensor, norm: torch.nn.LayerNorm, residual: Optional[Tensor] = None,
                     row_scale: Optional[Tensor] = None, alpha: Optional[Tensor] = None,
                     out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """``norm((x [+ residual]) * (1 + row_scale * alpha))`` in one launch (see include/salience_hip.h (6)).
    ``x`` may be a batch-strided view (e.g. one level's slice of ``[B,S,C]``); the result is contiguous."""
    if not x.is_cuda:
        raise RuntimeError("fused_layer_norm: HIP device tensors required; there is no CPU fallback")
    shape = x.shape
    xv, xbs, xrs = _bn_view(x)
    B, n, C = xv.shape
    rv, rbs, rrs = (None, 0, 0)
    if residual is not None:
        if residual.dtype != x.dtype or residual.shape != x.shape:
            raise RuntimeError("fused_layer_norm: residual must match x")
        rv, rbs, rrs = _bn_view(residual)
    if row_scale is not None:
        row_scale = row_scale.reshape(-1)
        if row_scale.dtype != torch.float32 or row_scale.numel() != B * n or not row_scale.is_contiguous():
            row_scale = row_scale.float().contiguous()
    w, b = norm.weight.detach(), norm.bias.detach()
    out_dtype = out_dtype or x.dtype
    out = torch.empty((B, n, C), dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
        code = _hip.lib().sdetr_layernorm(
            _hip.stream_ptr(), xv.data_ptr(), _hip.ptr(rv), _hip.dtype_code(x.dtype), xbs, xrs, rbs, rrs,
            _hip.ptr(row_scale), _hip.ptr(alpha), w.data_ptr(), b.data_ptr(), _hip.dtype_code(w.dtype),
            float(norm.eps), B, n, C, out.data_ptr(), _hip.dtype_code(out_dtype))
    _hip.check(code, "fused_layer_norm")
    return out.view(shape)


def column_mean(x: Tensor) -> Tensor:
    """Mean over dim 1 of a ``[B,n,C]`` fp32 tensor (may be a strided column slice) -> ``[B,1,C]``."""
    if not x.is_cuda:
        raise RuntimeError("column_mean: HIP device tensors required; there is no CPU fallback")
    if x.dtype != torch.float32 or x.dim() != 3 or x.stride(2) != 1:
        raise RuntimeError("column_mean: fp32 [B,n,C] with a contiguous last dim expected")
    B, n, C = x.shape
    out = torch.empty((B, 1, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        code = _hip.lib().sdetr_column_mean_f32(_hip.stream_ptr(), x.data_ptr(), x.stride(0), x.stride(1), B, n, C,
                                                out.data_ptr())
    _hip.check(code, "column_mean")
    return out
