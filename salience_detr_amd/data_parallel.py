"""Multi-GPU story of the hot path: independent images, one process per GPU.

The reference trains with HuggingFace ``accelerate`` -> ``DistributedDataParallel`` over NCCL
(``main.py:94-103,144``; ``util/engine.py:58``): data parallel only, the hot path itself contains no
collective.  Here:

* inference: ``shard_range`` gives every rank a contiguous slice of the images; ranks never talk;
* training step: every rank runs forward/backward on its own images and the gradients of the hot-path
  parameters (~38 MB fp32) are summed with ONE all-reduce over a flat buffer (``FlatGradAllReducer``).
  On MI355X ``backend="nccl"`` is RCCL over xGMI: 8 GPUs fully connected by 7 point-to-point links of
  ~153 GB/s, so a ring all-reduce is bound by one link (2*(7/8)*38 MB / 153 GB/s ~= 0.43 ms) while a
  direct reduce-scatter + all-gather drives all 7 links (~0.06 ms + latency).  With only 38 MB per step
  the right bucket is "everything at once": per-bucket latency (~20-30 us per collective), not bandwidth,
  is what DDP's default 25 MB buckets would multiply.  ``bucket_bytes`` is still configurable for overlap
  with a longer backward.
"""
from typing import Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous ``[start, stop)`` slice of ``num_items`` independent images for ``rank`` (sizes differ by
    at most one; empty when there are fewer images than ranks)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(num_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s parameters and buffers."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class FlatGradAllReducer:
    """Sum-all-reduce (and average) the gradients of ``params`` through flat buckets.

    Parameters are de-duplicated by identity (``encoder.enhance_mcsp`` IS ``encoder_class_head``).  A
    parameter without a gradient contributes zeros, so ranks may disagree on which parameters were used
    (the reference needs ``find_unused_parameters`` for that, ``main.py:101``).
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: Optional[int] = None, group=None):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = group
        first = self.params[0]
        self.dtype, self.device = first.dtype, first.device
        self.buckets: List[List[torch.nn.Parameter]] = [[]]
        size = 0
        for p in self.params:
            if p.dtype != self.dtype or p.device != self.device:
                raise ValueError("all parameters of one reducer must share dtype and device")
            nbytes = p.numel() * p.element_size()
            if bucket_bytes and self.buckets[-1] and size + nbytes > bucket_bytes:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nbytes
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=self.dtype, device=self.device)
                     for b in self.buckets]

    @property
    def num_bytes(self) -> int:
        return sum(f.numel() * f.element_size() for f in self.flat)

    def all_reduce(self, average: bool = True) -> None:
        """Pack grads -> all-reduce each bucket (async, then wait) -> unpack into ``p.grad``."""
        world = dist.get_world_size(self.group)
        works = []
        for bucket, flat in zip(self.buckets, self.flat):
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
                off += n
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        for bucket, flat in zip(self.buckets, self.flat):
            if average:
                flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n


class OverlappedGradReducer(FlatGradAllReducer):
    """``FlatGradAllReducer`` whose buckets are reduced WHILE backward is still running (the reference gets the
    same from ``DistributedDataParallel``'s bucket hooks, ``main.py:94-103``, ``util/engine.py:58``).

    Parameters are bucketed in REVERSE registration order -- roughly the order in which backward produces their
    gradients -- and every parameter carries a ``post_accumulate_grad`` hook: the hook copies the finished gradient
    into its slot of the flat bucket and, when the bucket's last gradient has arrived, starts that bucket's
    all-reduce asynchronously (on RCCL's own stream, so it overlaps the remaining backward kernels).  ``finish()``
    -- called after ``backward()`` -- starts the buckets that are still incomplete on this rank (parameters another
    rank used and this one did not contribute zeros), waits, averages and writes the results back to ``p.grad``.
    Every rank starts the buckets in the same order (bucket index), as collectives require: a bucket that completes
    early waits for its predecessors.

    Contract: ONE ``backward()`` with the hooks live per ``finish()``.  Gradient accumulation (the reference's
    ``--accumulate-steps``, ``util/engine.py:44`` ``accelerator.accumulate``) runs the first micro-steps under
    ``no_sync()`` -- the hooks do nothing, ``p.grad`` accumulates locally -- and the last one outside it: its hooks
    then pack the accumulated ``p.grad``.  A second backward WITHOUT ``no_sync()`` is detected (a hook fires for a
    parameter that is already packed): ``finish()`` then waits for whatever was launched and falls back to the
    pack-at-the-end reduction of the base class over the accumulated ``p.grad`` -- correct, just not overlapped.

    Construction is a COLLECTIVE with ``own_group=True`` (``dist.new_group()``: every rank of the default group must build
    its reducer at the same point; pass ``group=`` / ``own_group=False`` for a reducer on a subset of ranks).

    Collective order.  By default the reducer owns a process group of its own (``own_group=True``): other collectives
    issued during backward on the default group -- the neck's SyncBatchNorm statistics, ``_SyncBatchNormTrain.backward``
    -- interleave with the bucket all-reduces differently on ranks that did not use a parameter (they launch that bucket
    only in ``finish()``); on one communicator that is a mismatched collective order, on separate ones it is legal.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: Optional[int] = 8 << 20, group=None,
                 own_group: bool = True):
        ps = [p for p in params]
        if group is None and own_group and dist.is_available() and dist.is_initialized():
            group = dist.new_group()   # collective: every rank constructs its reducer at the same point
        super().__init__(reversed(ps), bucket_bytes=bucket_bytes, group=group)
        self._slot = {}
        for bi, bucket in enumerate(self.buckets):
            off = 0
            for p in bucket:
                self._slot[id(p)] = (bi, off)
                off += p.numel()
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._enabled = True
        self._reset()

    def _reset(self):
        self._arrived = [0] * len(self.buckets)
        self._filled = set()
        self._works = [None] * len(self.buckets)
        self._next_launch = 0
        self._repack = False

    def _launch_ready(self, force: bool = False):
        while self._next_launch < len(self.buckets):
            bi = self._next_launch
            if not force and self._arrived[bi] < len(self.buckets[bi]):
                return
            if self._arrived[bi] < len(self.buckets[bi]):   # finish(): parameters without a gradient on this rank
                off = 0
                for p in self.buckets[bi]:
                    if id(p) not in self._filled:
                        self.flat[bi][off:off + p.numel()].zero_()
                    off += p.numel()
            self._works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._next_launch += 1

    def no_sync(self):
        """Context manager for the non-final micro-steps of gradient accumulation: the hooks do nothing, gradients
        accumulate in ``p.grad`` (``DistributedDataParallel.no_sync`` / ``accelerator.accumulate`` semantics)."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                self.prev, reducer._enabled = reducer._enabled, False

            def __exit__(self, *exc):
                reducer._enabled = self.prev
                return False
        return _NoSync()

    def _on_grad(self, p: torch.nn.Parameter):
        if not self._enabled:
            return
        if id(p) in self._filled:
            # a second backward since the last finish(): the slot may already be on the wire.  Leave the buffers
            # alone; finish() reduces the accumulated p.grad after the launched collectives have drained.
            self._repack = True
            return
        bi, off = self._slot[id(p)]
        self.flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
        self._filled.add(id(p))
        self._arrived[bi] += 1
        if not self._repack:
            self._launch_ready()

    def finish(self, average: bool = True) -> None:
        """After ``backward()``: reduce what is left, wait, average, unpack into ``p.grad``; ready for the next step."""
        world = dist.get_world_size(self.group)
        # Whether ANY rank saw a second hooked backward decides the path for ALL ranks: the fallback issues one more
        # collective than the normal path, so a per-rank decision (data-dependent unused parameters in the second
        # backward) would leave the ranks with different collective sequences.  One 1-element all-reduce per
        # step, issued BEHIND the last bucket (ranks have launched different numbers of buckets when they get here: only
        # after the forced launches is the sequence the same everywhere).
        self._launch_ready(force=True)   # every rank has now issued every bucket exactly once
        flag = self.flat[0].new_tensor([1.0 if self._repack else 0.0])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if float(flag.item()) > 0.0:
            # more than one backward reached the hooks somewhere: drain the buckets (results discarded), then reduce the
            # accumulated p.grad the non-overlapped way
            for w in self._works:
                w.wait()
            self._reset()
            FlatGradAllReducer.all_reduce(self, average)
            return
        self._launch_ready(force=True)
        for w in self._works:
            w.wait()
        for bucket, flat in zip(self.buckets, self.flat):
            if average:
                flat.div_(world)
            off = 0
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        self._reset()

    def all_reduce(self, average: bool = True) -> None:   # same call site as the non-overlapped reducer
        self.finish(average)

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class StaticGradAllReducer:
    """Gradient all-reduce for a step whose forward + backward is REPLAYED as a captured hipGraph.

    Hook-driven bucket reducers issue their collectives from autograd hooks -- host code that a graph replay never
    runs.  Here the captured region ends with ``pack()``: one multi-tensor copy of the step's gradients (which live at
    fixed addresses inside the graph's memory pool) into ONE flat buffer, zero where this rank produced none.  After the
    replay the host issues a single all-reduce of that buffer (38 MB of hot-path gradients: one collective, the latency
    of a bucket chain is what xGMI's seven point-to-point links would multiply, see the module docstring) and points
    every ``p.grad`` at its slice, which is what the (eager, fused) optimizer step then reads.  Same result as
    ``FlatGradAllReducer``; world size 1 works without a process group.
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        if not self.params:
            raise ValueError("no trainable parameters")
        first = self.params[0]
        if any(p.dtype != first.dtype or p.device != first.device for p in self.params):
            raise ValueError("all parameters of one reducer must share dtype and device")
        self.group = group
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=first.dtype, device=first.device)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    @property
    def num_bytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def pack(self) -> None:
        """Last call of the captured region (after ``backward()``): gradients -> flat buffer.

        Per parameter: no gradient -> its slice is zeroed; a gradient in its own storage -> copied in; ``p.grad`` IS the
        slice (what ``all_reduce`` leaves behind, and what a following backward under ``zero_grad(set_to_none=False)`` or
        gradient accumulation adds into) -> left alone: the freshly accumulated values are already in place.  (Until round
        4 the whole buffer was zeroed first, which wiped exactly those)"""
        zero, dst, src = [], [], []
        for v, p in zip(self.views, self.params):
            if p.grad is None:
                zero.append(v)
            elif p.grad is not v and not (p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape
                                          and p.grad.stride() == v.stride()):
                dst.append(v)
                src.append(p.grad)
        if zero:
            torch._foreach_zero_(zero)
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce(self, average: bool = True) -> None:
        """After the replay: one all-reduce, then every ``p.grad`` is its slice of the reduced buffer."""
        world = 1
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
            if world > 1 or self.group is not None:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if average and world > 1:
            self.flat.div_(world)
        for p, v in zip(self.params, self.views):
            p.grad = v


class _SyncBatchNormTrain(torch.autograd.Function):
    """BatchNorm2d in training mode over ALL ranks' pixels (what ``nn.SyncBatchNorm`` gives the reference's neck under
    DDP, ``main.py:126-127``; single process: plain batch statistics).  One collective per direction instead of the
    framework's all_gather + all_reduce pair: forward all-reduces ``[sum(x - s), sum((x - s)^2), count]`` (2C+1
    floats), backward all-reduces ``[sum(dy), sum(dy * xhat)]`` (2C floats).  ``s`` is a per-channel shift that is
    identical on every rank (the running mean: updated from global statistics only, so replicas agree): the variance
    ``E[(x-s)^2] - E[x-s]^2`` then cancels only as far as the batch mean has moved from the running mean, not as far as
    it is from zero (torch's SyncBatchNorm merges per-rank mean / M2 instead, which needs the gathered per-rank
    statistics).  Running statistics are updated in place with the UNBIASED variance of the global batch
    (``nn.BatchNorm2d`` semantics); ``weight`` / ``bias`` / running statistics may be ``None``."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        C = x.shape[1]
        xf = x.float()
        red = (0, 2, 3)
        shift = running_mean.detach().float() if running_mean is not None else xf.new_zeros(C)
        xs = xf - shift.view(1, C, 1, 1)
        stats = torch.cat([xs.sum(red), (xs * xs).sum(red), xf.new_tensor([xf.numel() / C])])
        world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
        n = stats[-1]
        dmean = stats[:C] / n                                              # E[x - s]
        var = (stats[C:2 * C] / n - dmean * dmean).clamp_min_(0.0)         # biased: what normalises
        mean = dmean + shift
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None and running_var is not None:
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(momentum * mean.to(running_mean.dtype))
                running_var.mul_(1 - momentum).add_(momentum * (var * (n / (n - 1).clamp_min(1.0))).to(running_var.dtype))
        xhat = (xs - dmean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
        ctx.save_for_backward(xhat, weight, invstd, n)
        ctx.group, ctx.world, ctx.has_bias = group, world, bias is not None
        y = xhat
        if weight is not None:
            y = y * weight.float().view(1, C, 1, 1)
        if bias is not None:
            y = y + bias.float().view(1, C, 1, 1)
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, invstd, n = ctx.saved_tensors
        C = dy.shape[1]
        dyf = dy.float()
        red = (0, 2, 3)
        sum_dy, sum_dy_xhat = dyf.sum(red), (dyf * xhat).sum(red)
        both = torch.cat([sum_dy, sum_dy_xhat])
        if ctx.world > 1:
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=ctx.group)
        g_mean, g_proj = both[:C] / n, both[C:] / n
        scale = invstd if weight is None else weight.float() * invstd
        dx = (dyf - g_mean.view(1, C, 1, 1) - xhat * g_proj.view(1, C, 1, 1)) * scale.view(1, C, 1, 1)
        # parameter gradients stay LOCAL sums: the gradient all-reduce of the step adds the ranks' contributions
        gw = sum_dy_xhat.to(weight.dtype) if weight is not None else None
        gb = sum_dy.to(dy.dtype if weight is None else weight.dtype) if ctx.has_bias else None
        return dx.to(dy.dtype), gw, gb, None, None, None, None, None


def sync_batch_norm_train(x: torch.Tensor, bn: torch.nn.modules.batchnorm._BatchNorm, group=None) -> torch.Tensor:
    """``bn`` (an ``nn.BatchNorm2d`` parameter holder) applied to NCHW ``x`` with batch statistics over every rank of
    ``group`` (all ranks of the default group when a process group exists, else this process alone).  Follows
    ``nn.BatchNorm2d``'s conventions: ``momentum=None`` is the cumulative moving average (factor ``1 / num_batches``),
    ``affine=False`` / ``track_running_stats=False`` holders have no weight / running statistics."""
    factor = 0.0
    if bn.running_mean is not None and getattr(bn, "num_batches_tracked", None) is not None:
        bn.num_batches_tracked.add_(1)
        # (momentum=None reads the counter: one host read per call, only in that rarely used mode)
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else bn.momentum
    elif bn.momentum is not None:
        factor = bn.momentum
    return _SyncBatchNormTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, factor, group)
