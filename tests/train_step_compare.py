"""Shared by the full-size training-step tests (CPU: oracle autograd against the reference fixture; GPU: the HIP step
against both): how a gradient is compared, and the fixture's sub-sampling (tests/golden/make_golden.py `sub`)."""
import torch


def sub(g: torch.Tensor) -> torch.Tensor:
    """A gradient as the fixture stores it: whole when small, every 4th row / column of the big matrices."""
    g = g.detach()
    if g.numel() > 4096 and g.dim() >= 2:
        g = g[::4, ::4]
    return g


def compare(got: torch.Tensor, ref: torch.Tensor, name: str, scale: float = None):
    """(worst error / the gradient's scale, rows left out).  The ReLU's gate is discontinuous: a hidden unit whose
    pre-activation is within rounding of zero for some token (expected: ~1e-6 of the T x 2048 of them, a handful per layer)
    is on in one run and off in the other -- GPU against host, oracle against reference, or two GPU runs whose split
    reductions add in another order -- and that token's whole contribution dh[t, j] * x[t] appears in / vanishes from row j
    of linear1's gradient: a few percent of a row that sums ~1000 active tokens.  Rows of linear1's gradients are therefore
    compared one by one and the few beyond 1e-2 counted instead of bounded."""
    if scale is None:
        scale = ref.abs().max().item()
    assert scale > 0.0, name
    err = (got - ref).abs() / scale
    off = 0
    if ".linear1." in name:
        per_row = err.reshape(err.shape[0], -1).max(1)[0]
        off = int((per_row > 1e-2).sum())
        err = per_row[per_row <= 1e-2] if off else per_row
    return err.max().item(), off


def loss_fn(memory, score_maps, w, mean=lambda t: t.mean()):
    """`train_full_loss` of tests/golden/make_golden.py."""
    return mean(memory * w) * 100.0 + sum((s * s).mean() for s in score_maps)
