"""fp32 ``nn.Linear`` forward / backward on the bf16 matrix cores at fp32 accuracy (``csrc/gemm_x3.hip``).

The training step's Linear layers (``models/bricks/salience_transformer.py:347-351`` FFN, ``:366-379`` attention
projections, ``ms_deform_attn.py:312-331`` value / offset / weight / output projections) are fp32 GEMMs with K = 256
or 2048; autograd sends them to the library's fp32 GEMM, which runs at ~56 TFLOP/s on these shapes.
``sdetr_gemm_x3_f32`` splits every fp32 operand exactly into three bf16 terms on its way into LDS and takes each
product as six bf16 MFMAs -- the result is an fp32 product (error below 2^-24 of each term), the rate is that of the
bf16 matrix cores divided by six.  ``use_x3_linear_(model)`` switches a model's ``nn.Linear`` modules to it in place
(same parameters, same ``state_dict`` keys); inputs that do not meet the kernel's alignment rules, CPU tensors and
non-fp32 dtypes keep going through ``F.linear``.
"""
import contextlib
from typing import Optional

import torch
from torch import Tensor, nn
from torch.autograd import Function
from torch.nn import functional as F

from . import _hip
from .zero_arena import zeros as _zeros


EPI_NONE, EPI_RELU, EPI_GATE = 0, 1, 2   # SDETR_GEMM_EPI_* (include/salience_hip.h)


def gemm_x3(a: Tensor, a_kmajor: bool, b: Tensor, b_kmajor: bool, M: int, N: int, K: int,
            bias: Optional[Tensor] = None, reduction_splits: int = 1, out: Optional[Tensor] = None,
            a_row_sum: Optional[Tensor] = None, epilogue: int = EPI_NONE, gate: Optional[Tensor] = None) -> Tensor:
    """``C[M,N] = sum_k A(m,k) B(n,k) (+ bias[n])`` with ``A(m,k) = a[m,k]`` (k-major) or ``a[k,m]``; ``b`` likewise
    (``include/salience_hip.h``).  ``a`` / ``b`` are 2-d fp32 HIP tensors with a contiguous last dimension.
    ``a_row_sum`` (fp32 [M], zeroed by the caller, reduction-major ``a`` only) receives ``sum_k A(m,k)``.
    ``epilogue``: ``EPI_RELU`` clamps ``C`` at zero, ``EPI_GATE`` zeroes it where ``gate`` (fp32 ``[M, N]``) is not
    positive -- unsplit reductions only."""
    _hip.require_device("gemm_x3", a=a, b=b, bias=bias, gate=gate)
    if epilogue == EPI_GATE and (gate is None or gate.dtype != torch.float32 or tuple(gate.shape) != (M, N)
                                 or gate.stride(1) != 1):
        raise RuntimeError("gemm_x3: the gate epilogue needs an fp32 [M, N] gate with a contiguous last dimension")
    for t, what in ((a, "a"), (b, "b")):
        if t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1:
            raise RuntimeError(f"gemm_x3: {what} must be a 2-d fp32 tensor with a contiguous last dimension")
    if tuple(a.shape) != ((M, K) if a_kmajor else (K, M)) or tuple(b.shape) != ((N, K) if b_kmajor else (K, N)):
        raise RuntimeError("gemm_x3: operand shapes do not match (M, N, K) and the layouts")
    if out is None:
        c = (_zeros if reduction_splits > 1 else torch.empty)((M, N), dtype=torch.float32, device=a.device)
    else:   # a contiguous fp32 buffer of M * N elements in any shape (zeroed by the caller for a split reduction)
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != M * N or out.device != a.device:
            raise RuntimeError("gemm_x3: out must be a contiguous fp32 tensor of M * N elements on the operands' device")
        c = out.view(M, N)
    with torch.cuda.device(a.device):
        code = _hip.lib().sdetr_gemm_x3_epilogue_f32(
            _hip.stream_ptr(), a.data_ptr(), a.stride(0), int(a_kmajor), b.data_ptr(), b.stride(0), int(b_kmajor),
            c.data_ptr(), c.stride(0), M, N, K, _hip.ptr(bias), int(reduction_splits), _hip.ptr(a_row_sum),
            int(epilogue), _hip.ptr(gate) if epilogue == EPI_GATE else None, gate.stride(0) if epilogue == EPI_GATE else 0)
    _hip.check(code, "gemm_x3")
    return c if out is None else out


def presplit(w: Tensor, transpose: bool = False) -> Tensor:
    """The three bf16 planes ``[3, rows, cols]`` (or ``[3, cols, rows]`` transposed) of an fp32 matrix: the pre-split
    form ``gemm_x3`` takes for ``b`` (``b_kmajor = 2``) -- a weight is split once per call instead of in every
    workgroup that reads it."""
    _hip.require_device("presplit", w=w)
    if w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
        raise RuntimeError("presplit: 2-d fp32 tensor with a contiguous last dimension expected")
    R, C = w.shape
    out = torch.empty((3, C, R) if transpose else (3, R, C), dtype=torch.bfloat16, device=w.device)
    with torch.cuda.device(w.device):
        code = _hip.lib().sdetr_gemm_x3_presplit(_hip.stream_ptr(), w.data_ptr(), w.stride(0), R, C, int(transpose),
                                                 out.data_ptr())
    _hip.check(code, "presplit")
    return out


@contextlib.contextmanager
def pinned_generation(generation: int):
    """Pins the kernel generation of this thread's ``gemm_x3`` calls (1 = 128 x 128 tiles, 2 = 256 x 128 tiles;
    ``sdetr_gemm_x3_generation``) for the duration of the block: the parity tests run every shape on both."""
    before = _hip.lib().sdetr_gemm_x3_generation(int(generation))
    try:
        yield
    finally:
        _hip.lib().sdetr_gemm_x3_generation(before)


def gemm_x3_presplit_b(a: Tensor, a_kmajor: bool, b_planes: Tensor, M: int, N: int, K: int,
                       bias: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """``gemm_x3`` with ``b`` given as ``presplit`` planes ``[3, N, K]`` (K a multiple of 8)."""
    _hip.require_device("gemm_x3", a=a, b=b_planes, bias=bias)
    if (a.dtype != torch.float32 or a.dim() != 2 or a.stride(1) != 1 or b_planes.dtype != torch.bfloat16
            or tuple(b_planes.shape) != (3, N, K) or not b_planes.is_contiguous()
            or tuple(a.shape) != ((M, K) if a_kmajor else (K, M))):
        raise RuntimeError("gemm_x3_presplit_b: fp32 a, contiguous bf16 planes [3, N, K] expected")
    c = torch.empty((M, N), dtype=torch.float32, device=a.device) if out is None else out.view(M, N)
    with torch.cuda.device(a.device):
        code = _hip.lib().sdetr_gemm_x3_f32(_hip.stream_ptr(), a.data_ptr(), a.stride(0), int(a_kmajor),
                                            b_planes.data_ptr(), K, 2, c.data_ptr(), c.stride(0), M, N, K,
                                            _hip.ptr(bias), 1, None)
    _hip.check(code, "gemm_x3")
    return c if out is None else out


def x3_linear_applies(x: Tensor, weight: Tensor, bias: Optional[Tensor]) -> bool:
    """fp32 HIP tensors whose sizes meet the kernel's alignment rule (in / out features multiples of 4)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.dim() == 2
            and weight.is_contiguous() and weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0
            and x.shape[-1] == weight.shape[1] and x.numel() > 0
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())))


def _weight_grad_splits(T: int, N: int, K: int) -> int:
    """Slices of the token dimension for ``dw = dy^T x``: its output is only a few tiles, so the reduction over the
    tokens is what has to fill the chip (measured: benchmarks/gemm_x3_dw_splits_probe.py -- 256 x 256 at 22 726 tokens:
    50-52 us at 88-128 slices of the 256 x 128-tile generation, 61-63 on the 128 x 128 tiles, 91 in the library)."""
    if T >= 1024 and N >= 128:   # the library's 256 x 128-tile generation takes it: one workgroup per CU
        tiles = ((N + 255) // 256) * ((K + 127) // 128)
        return max(1, min(T // 256, 256 // tiles))
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    return max(1, min((T + 255) // 256, (512 + tiles - 1) // tiles))


# Which of a Linear layer's three products take the x3 kernel (the others go to the library's fp32 GEMM).  Measured on
# MI355X (benchmarks/gemm_x3_bench.py, profiles/r03_gemm_x3.json): the weight gradient dw = dy^T x always (1.3-2x the
# library, whose few output tiles are not split over the token dimension); y = x w^T and dx = dy w where an output or
# reduction dimension is long enough for the 256 x 128-tile generation (``_x3_wide``: 146-200 us against the library's
# 189-275 at 22 726 tokens) -- the weight is read as it lies (k-major for y, reduction-major for dx) and split on its way
# into LDS; the pre-split form (``presplit`` + ``gemm_x3_presplit_b``) is kept for callers that reuse one weight often.
X3_FORWARD, X3_DX, X3_DW = True, True, True
# (module attributes: the tests lower them to route everything here)
X3_WIDE_FEATURES = 2048         # a product counts as wide / long from this many output / reduction features
X3_WIDE_OUT_ROWS = 9000         # ... and goes to the x3 kernel from this many tokens when its OUTPUT is wide
X3_LONG_REDUCTION_ROWS = 4000   # ... from this many when its REDUCTION is long (cut into slices until the tiles fill the chip)


def _x3_wide(T: int, n_out: int, k_red: int) -> bool:
    """The forward / input-gradient products that go to the x3 kernel (round 3: its 256 x 128 tile generation; measured
    against the library's fp32 GEMM, benchmarks/gemm_x3_bench.py): a 2048-wide output from 9000 tokens (127 vs 159 us at
    13 634, 200 vs 275 at 22 726), a 2048-long reduction from 4000 tokens (131 vs 151 at 18 180; below that with the
    reduction cut in slices, ``_reduction_splits``); narrower products stay with the library."""
    if n_out >= X3_WIDE_FEATURES and T >= X3_WIDE_OUT_ROWS:
        return True
    return k_red >= X3_WIDE_FEATURES and T >= X3_LONG_REDUCTION_ROWS


def _reduction_splits(T: int, n_out: int, k_red: int) -> int:
    """Slices of a LONG reduction for y = x w^T / dx = dy w: the 256 x 128 tiles of a narrow output do not fill the chip
    below ~32 000 tokens (13 634 x 256: 108 tiles on 256 CUs), so the reduction is cut until they do -- 100 vs 130 us at
    13 634 tokens, 79 vs 132 at 9090 (the library: 128 / 103); the slices add into a zeroed output with fp32 atomics."""
    if k_red < 1024:
        return 1
    tiles = ((T + 255) // 256) * ((n_out + 127) // 128)
    return max(1, min(256 // max(tiles, 1), k_red // 256))


class _LinearX3(Function):
    """``x`` [..., K] contiguous.  The output is allocated in its final shape and the products write into it through
    2-d views: a view of a custom Function's output that is then modified in place (``nn.ReLU(inplace=True)`` follows
    ``linear1``) makes autograd rebase the history on a CopySlices node, whose backward copies the gradient of the
    [T, 2048] hidden state three times -- 0.9 ms of the training step; an output that IS the base has no such node."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = x.shape[-1]
        x2 = x.view(-1, K)
        T, N = x2.shape[0], weight.shape[0]
        use_x3 = X3_FORWARD and K % 8 == 0 and _x3_wide(T, N, K)
        splits = _reduction_splits(T, N, K) if use_x3 else 1
        y = (_zeros if splits > 1 else torch.empty)(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        y2 = y.view(T, N)
        if use_x3:   # y = x w^T (the weight is split on its way into LDS)
            gemm_x3(x2, True, weight, True, T, N, K, bias=bias, reduction_splits=splits, out=y2)
        elif bias is not None:
            torch.addmm(bias, x2, weight.t(), out=y2)
        else:
            torch.mm(x2, weight.t(), out=y2)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        T, K = x2.shape
        N = weight.shape[0]
        g2 = (gy if gy.is_contiguous() else gy.contiguous()).view(T, N)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:                                                    # dx = dy w
            if X3_DX and N % 8 == 0 and _x3_wide(T, K, N):
                # dx = dy w (the weight read reduction-major: no transposed split pass)
                sp = _reduction_splits(T, K, N)
                gx = gemm_x3(g2, True, weight, False, T, K, N, reduction_splits=sp,
                             out=_zeros((T, K), dtype=g2.dtype, device=g2.device) if sp > 1 else None)
            else:
                gx = g2 @ weight
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:                                                    # dw = dy^T x (+ db = sum_t dy)
            if X3_DW:
                splits = _weight_grad_splits(T, N, K)
                out = None
                if want_gb:   # the kernel has dy's tiles in registers anyway: the bias gradient is their row sums
                    # (one zero fill for both gradients: the split reduction adds into dw as the row sums add into db)
                    buf = _zeros(N * K + N, dtype=torch.float32, device=g2.device)
                    out, gb = buf[:N * K].view(N, K), buf[N * K:]
                elif splits > 1:
                    out = _zeros((N, K), dtype=torch.float32, device=g2.device)
                gw = gemm_x3(g2, False, x2, False, N, K, T, reduction_splits=splits, out=out, a_row_sum=gb)
                if out is not None:
                    gw = out
            else:
                gw = g2.t() @ x2
        if want_gb and gb is None:
            gb = g2.sum(0)
        return (None if gx is None else gx.view(ctx.x_shape)), gw, gb


def _weight_and_bias_grad(g2: Tensor, x2: Tensor, want_gb: bool):
    """``dw = g2^T x2`` (+ ``db = sum_t g2``) through the x3 kernel: one zero fill for both, the bias gradient from the
    row sums of the tiles the kernel holds anyway."""
    T, N = g2.shape
    K = x2.shape[1]
    splits = _weight_grad_splits(T, N, K)
    out = gb = None
    if want_gb:
        buf = _zeros(N * K + N, dtype=torch.float32, device=g2.device)
        out, gb = buf[:N * K].view(N, K), buf[N * K:]
    elif splits > 1:
        out = _zeros((N, K), dtype=torch.float32, device=g2.device)
    gw = gemm_x3(g2, False, x2, False, N, K, T, reduction_splits=splits, out=out, a_row_sum=gb)
    return (gw if out is None else out), gb


class _FfnX3(Function):
    """``linear2(relu(linear1(x)))`` (``models/bricks/salience_transformer.py:347-351`` without its dropout) as ONE autograd
    node: the ReLU is the epilogue of the first product forward and of linear2's input-gradient product backward
    (``EPI_RELU`` / ``EPI_GATE``), so the two elementwise passes over the ``[T, F]`` hidden state and the in-place
    bookkeeping autograd does for ``nn.ReLU(inplace=True)`` disappear.  Products that the shape routing of ``_LinearX3``
    leaves with the library keep its GEMM, followed by the elementwise pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        K = x.shape[-1]
        x2 = x.view(-1, K)
        T, Fh, N = x2.shape[0], w1.shape[0], w2.shape[0]
        if X3_FORWARD and K % 8 == 0 and _x3_wide(T, Fh, K) and _reduction_splits(T, Fh, K) == 1:
            h = gemm_x3(x2, True, w1, True, T, Fh, K, bias=b1, epilogue=EPI_RELU)
        else:
            h = torch.addmm(b1, x2, w1.t()) if b1 is not None else torch.mm(x2, w1.t())
            h.clamp_min_(0.0)
        use_x3 = X3_FORWARD and Fh % 8 == 0 and _x3_wide(T, N, Fh)
        splits = _reduction_splits(T, N, Fh) if use_x3 else 1
        y = (_zeros if splits > 1 else torch.empty)(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
        y2 = y.view(T, N)
        if use_x3:
            gemm_x3(h, True, w2, True, T, N, Fh, bias=b2, reduction_splits=splits, out=y2)
        elif b2 is not None:
            torch.addmm(b2, h, w2.t(), out=y2)
        else:
            torch.mm(h, w2.t(), out=y2)
        ctx.save_for_backward(x2, h, w1, w2)
        ctx.has_bias = (b1 is not None, b2 is not None)
        ctx.x_shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, h, w1, w2 = ctx.saved_tensors
        T, K = x2.shape
        Fh, N = w1.shape[0], w2.shape[0]
        g2 = (gy if gy.is_contiguous() else gy.contiguous()).view(T, N)
        need = ctx.needs_input_grad
        gx = gw1 = gb1 = gw2 = gb2 = None
        if need[3]:
            gw2, gb2 = _weight_and_bias_grad(g2, h, ctx.has_bias[1] and need[4])
        elif ctx.has_bias[1] and need[4]:
            gb2 = g2.sum(0)
        if need[0] or need[1] or need[2]:
            # dh = (dy w2) * (h > 0)
            if X3_DX and N % 8 == 0 and _x3_wide(T, Fh, N) and _reduction_splits(T, Fh, N) == 1:
                dh = gemm_x3(g2, True, w2, False, T, Fh, N, epilogue=EPI_GATE, gate=h)
            else:
                dh = torch.ops.aten.threshold_backward(g2 @ w2, h, 0.0)
            if need[1]:
                gw1, gb1 = _weight_and_bias_grad(dh, x2, ctx.has_bias[0] and need[2])
            elif ctx.has_bias[0] and need[2]:
                gb1 = dh.sum(0)
            if need[0]:
                if X3_DX and Fh % 8 == 0 and _x3_wide(T, K, Fh):
                    sp = _reduction_splits(T, K, Fh)
                    gx = gemm_x3(dh, True, w1, False, T, K, Fh, reduction_splits=sp,
                                 out=_zeros((T, K), dtype=dh.dtype, device=dh.device) if sp > 1 else None)
                else:
                    gx = dh @ w1
                gx = gx.view(ctx.x_shape)
        return gx, gw1, gb1, gw2, gb2


def x3_ffn_applies(x: Tensor, linear1: nn.Linear, linear2: nn.Linear) -> bool:
    """Both layers meet ``x3_linear_applies`` (fp32 HIP tensors, feature counts multiples of 4) and chain."""
    w2, b2 = linear2.weight, linear2.bias
    return (X3_DW and x3_linear_applies(x, linear1.weight, linear1.bias) and w2.is_cuda and w2.dtype == torch.float32
            and w2.dim() == 2 and w2.is_contiguous() and w2.shape[1] == linear1.weight.shape[0] and w2.shape[0] % 4 == 0
            and w2.shape[1] % 4 == 0 and (b2 is None or (b2.dtype == torch.float32 and b2.is_contiguous())))


def x3_ffn(x: Tensor, linear1: nn.Linear, linear2: nn.Linear) -> Tensor:
    """``linear2(relu(linear1(x)))`` through ``_FfnX3`` (differentiable; the caller checks ``x3_ffn_applies``)."""
    return _FfnX3.apply(x if x.is_contiguous() else x.contiguous(), linear1.weight, linear1.bias, linear2.weight,
                        linear2.bias)


def x3_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """``F.linear`` with the fp32 products on the bf16 matrix cores (differentiable)."""
    if not x3_linear_applies(x, weight, bias):
        return F.linear(x, weight, bias)
    return _LinearX3.apply(x if x.is_contiguous() else x.contiguous(), weight, bias)


class X3Linear(nn.Linear):
    """``nn.Linear`` whose fp32 HIP forward / backward run through ``x3_linear``."""

    def forward(self, input: Tensor) -> Tensor:   # noqa: A002 (nn.Linear's own argument name)
        return x3_linear(input, self.weight, self.bias)


def use_x3_linear_(model: nn.Module) -> int:
    """Switch every plain ``nn.Linear`` of ``model`` to ``X3Linear`` in place (parameters and ``state_dict`` keys
    unchanged; ``NonDynamicallyQuantizableLinear`` -- the out_proj of ``nn.MultiheadAttention``, which is called
    through ``F.multi_head_attention_forward`` and never through its own ``forward`` -- is left alone).  Returns the
    number of modules switched."""
    n = 0
    for m in model.modules():
        if type(m) is nn.Linear:
            m.__class__ = X3Linear
            n += 1
        elif hasattr(m, "x3_projections"):   # modules that call F.linear on parameters of their own (attention in / out
            m.x3_projections = True          # projections): they route those calls through x3_linear when this is set
    return n
