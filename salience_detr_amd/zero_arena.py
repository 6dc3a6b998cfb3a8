"""One zero fill per training step instead of one per small workspace.

The backward of the training step allocates ~100 small zero-initialised buffers per step -- the weight / bias gradient
pairs the split-reduction GEMMs accumulate into (``linear_x3``), the gamma / beta sums of every LayerNorm backward
(``layer_norm_train``) -- and each ``torch.zeros`` is a fill launch of ~3.7 us whatever its size (0.4 ms of a 16 ms step,
``profiles/r04_train_ops.txt``).  Inside ``with zero_arena(device):`` those sites take slices of ONE persistent buffer that
``begin_step()`` clears with one fill.

Contract: a slice stays valid (and keeps whatever was accumulated into it -- parameter gradients alias it) until the next
``begin_step()``; consume the gradients (optimizer step, all-reduce pack) before that.  Without an active arena, or for
requests that do not fit, ``zeros`` is ``torch.zeros``.  The buffer lives outside any captured graph's pool, so a replayed
step reuses the same addresses."""
import contextlib
from typing import Optional, Sequence

import torch

# Process-wide, not thread-local: the sites that ask for zeros run in autograd's backward thread, not in the thread that
# entered the context (a thread-local arena was never seen by them: measured no gain, every request fell back).
_active = [None]


class ZeroArena:
    def __init__(self, device, capacity_bytes: int = 96 << 20, max_item_bytes: int = 4 << 20):
        self.buf = torch.zeros(capacity_bytes // 4, dtype=torch.float32, device=device)
        self.max_item = max_item_bytes // 4
        self.offset = 0
        self.high_water = 0
        self.misses = 0

    def begin_step(self) -> None:
        """Clear what the previous step used (one fill) and start handing out slices from the front again."""
        used = max(self.offset, self.high_water)
        if used:
            self.buf[:used].zero_()
        self.high_water = max(self.high_water, self.offset)
        self.offset = 0

    def take(self, numel: int) -> Optional[torch.Tensor]:
        n = (numel + 63) & ~63   # 256-byte aligned slices: the GEMM kernels want 16-byte aligned operands
        if numel > self.max_item or self.offset + n > self.buf.numel():
            self.misses += 1
            return None
        t = self.buf[self.offset:self.offset + numel]
        self.offset += n
        return t


def active() -> Optional[ZeroArena]:
    return _active[0]


@contextlib.contextmanager
def zero_arena(arena: ZeroArena):
    prev = _active[0]
    _active[0] = arena
    try:
        yield arena
    finally:
        _active[0] = prev


def zeros(shape: Sequence[int], device, dtype=torch.float32) -> torch.Tensor:
    """``torch.zeros(shape)`` -- from the active arena when there is one on this device and the request fits."""
    a = active()
    if a is not None and dtype == torch.float32 and a.buf.device == torch.device(device):
        numel = 1
        for d in shape:
            numel *= int(d)
        t = a.take(numel) if numel > 0 else None
        if t is not None:
            return t.view(tuple(shape))
    return torch.zeros(tuple(shape), dtype=dtype, device=device)
