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
