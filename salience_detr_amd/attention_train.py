"""Dense self-attention over a few hundred rows for the training step, one launch forward and two backward
(``csrc/attention_train.hip``): the core of the encoder layer's ``nn.MultiheadAttention`` over its top-300 rows
(models/bricks/salience_transformer.py:371-376) between the in- and the out-projection.

``attention_qk_v(qk, v, num_heads)``: ``qk`` [B,N,2E] = the q and k projections of a row side by side (one GEMM over the
``q = k`` input), ``v`` [B,N,E] the value projection; returns the heads' outputs concatenated [B,N,E].  fp32 HIP tensors
with 32-channel heads and at most ``sdetr_attention_train_max_rows()`` rows; ``applies`` says when.
"""
import math

import torch
from torch import Tensor
from torch.autograd import Function

from . import _hip


def applies(qk: Tensor, v: Tensor, num_heads: int) -> bool:
    return (qk.is_cuda and qk.dtype == torch.float32 and v.dtype == torch.float32 and qk.dim() == 3 and v.dim() == 3
            and qk.shape[-1] == 2 * v.shape[-1] and v.shape[-1] == 32 * num_heads and qk.shape[:2] == v.shape[:2]
            and 0 < qk.shape[1] <= _hip.lib().sdetr_attention_train_max_rows())


class _AttentionQKV(Function):
    @staticmethod
    def forward(ctx, qk, v, num_heads):
        qk, v = qk.contiguous(), v.contiguous()
        B, N, E = v.shape
        out = torch.empty_like(v)
        lse = torch.empty((B, num_heads, N), dtype=torch.float32, device=v.device)
        scale = 1.0 / math.sqrt(E // num_heads)
        with torch.cuda.device(v.device):
            code = _hip.lib().sdetr_attention_train_forward_f32(
                _hip.stream_ptr(), qk.data_ptr(), N * 2 * E, 2 * E, qk.data_ptr() + 4 * E, N * 2 * E, 2 * E, v.data_ptr(), N * E, E,
                B, num_heads, N, E // num_heads, scale, out.data_ptr(), lse.data_ptr())
        _hip.check(code, "attention_train_forward")
        ctx.save_for_backward(qk, v, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, grad_out):
        qk, v, out, lse = ctx.saved_tensors
        B, N, E = v.shape
        H = ctx.num_heads
        go = grad_out.contiguous()
        gqk, gv = torch.empty_like(qk), torch.empty_like(v)
        with torch.cuda.device(v.device):
            code = _hip.lib().sdetr_attention_train_backward_f32(
                _hip.stream_ptr(), qk.data_ptr(), N * 2 * E, 2 * E, qk.data_ptr() + 4 * E, N * 2 * E, 2 * E, v.data_ptr(), N * E, E,
                B, H, N, E // H, 1.0 / math.sqrt(E // H), out.data_ptr(), lse.data_ptr(), go.data_ptr(), gqk.data_ptr(),
                gqk.data_ptr() + 4 * E, gv.data_ptr())
        _hip.check(code, "attention_train_backward")
        return gqk, gv, None


def attention_qk_v(qk: Tensor, v: Tensor, num_heads: int) -> Tensor:
    return _AttentionQKV.apply(qk, v, num_heads)
