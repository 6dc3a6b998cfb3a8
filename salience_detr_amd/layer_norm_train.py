"""``norm(x + residual)`` of the training step as ONE launch each way (``csrc/layer_norm_train.hip``).

The encoder / decoder layers end every sub-block with ``norm(query + dropout(sublayer(query)))``
(models/bricks/salience_transformer.py:347-351, 377-378, 390-391, 571-588).  Under autograd the framework runs an add
and a LayerNorm kernel forward and a grad-input kernel, two gamma / beta reductions and an add backward; the op below
is one kernel forward (sum, normalised output, row statistics) and one backward (grad of the sum = grad of both
addends, gamma / beta gradients by atomics).  fp32 HIP tensors with 64 / 128 / 256 / 512 channels; everything else --
and a non-zero dropout in training mode, which the caller applies itself -- goes through ``torch`` ops.
"""
from typing import Optional

import torch
from torch import Tensor, nn
from torch.autograd import Function

from . import _hip, zero_arena


def applies(x: Tensor, norm: nn.LayerNorm, residual: Optional[Tensor] = None) -> bool:
    C = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine
            and norm.bias is not None and tuple(norm.normalized_shape) == (C,) and norm.weight.dtype == torch.float32
            and (residual is None or (residual.shape == x.shape and residual.dtype == x.dtype and residual.is_cuda))
            and x.numel() > 0 and bool(_hip.lib().sdetr_layer_norm_train_supported(C)))


class _AddLayerNorm(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps):
        C = x.shape[-1]
        x2 = x.contiguous().view(-1, C)
        r2 = None if residual is None else residual.contiguous().view(-1, C)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        z = torch.empty_like(x2) if r2 is not None else None
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        w, b = weight.contiguous(), bias.contiguous()
        with torch.cuda.device(x.device):
            code = _hip.lib().sdetr_layer_norm_train_forward_f32(
                _hip.stream_ptr(), x2.data_ptr(), _hip.ptr(r2), w.data_ptr(), b.data_ptr(), float(eps), rows, C,
                _hip.ptr(z), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr())
        _hip.check(code, "layer_norm_train_forward")
        ctx.save_for_backward(x2 if z is None else z, stats, w)
        ctx.has_residual = residual is not None
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        z, stats, w = ctx.saved_tensors
        rows, C = z.shape
        g2 = gy.contiguous().view(rows, C)
        dz = torch.empty_like(z)
        dwb = zero_arena.zeros((2, C), dtype=torch.float32, device=z.device)
        with torch.cuda.device(z.device):
            code = _hip.lib().sdetr_layer_norm_train_backward_f32(
                _hip.stream_ptr(), g2.data_ptr(), z.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), w.data_ptr(),
                rows, C, dz.data_ptr(), dwb[0].data_ptr(), dwb[1].data_ptr())
        _hip.check(code, "layer_norm_train_backward")
        dz = dz.view(ctx.shape)
        return dz, (dz if ctx.has_residual else None), dwb[0], dwb[1], None


def add_layer_norm(x: Tensor, norm: nn.LayerNorm, residual: Optional[Tensor] = None) -> Tensor:
    """``norm(x + residual)`` (``norm(x)`` without a residual), differentiable."""
    if not applies(x, norm, residual):
        return norm(x if residual is None else x + residual)
    return _AddLayerNorm.apply(x, residual, norm.weight, norm.bias, norm.eps)
