"""Several batches of the hot path in flight on one GPU: one captured hipGraph, one stream and one set of static
input / output tensors per *lane*.

One batch through the path is a dependent chain of ~70 launches, most of which cover a fraction of the chip for 5-30 us
(the 300-row attention, the top-k launches, the coarse pyramid levels): run strictly one after the other they leave CUs
idle that an independent batch can use.  The data path has nothing to exchange between batches, so a server with
queued requests keeps ``lanes`` of them going side by side -- measured on MI355X (``bench.py``, ResNet50 800x1333,
batch 2 per lane): 1.28 ms per batch one at a time, 0.89 with two lanes, 0.79 with three (2510 images/s), no further
gain beyond (the resident MSDA kernel and the feed-forward kernel take a whole CU each).  Outputs are bit-identical
to the lanes run alone: the lanes share the module's parameters and packed operands, all read-only, and nothing else.

    lanes = GraphLanes(lambda f, m, p: model(f, m, p, image_sizes=sizes, canvas=canvas)[0], (feats, masks, pos), lanes=3)
    lane = lanes.submit((feats_i, masks_i, pos_i))     # copy into the lane's inputs + replay, both on its stream
    ...
    lane.wait()                                        # the current stream waits for that replay
    use(lane.outputs)

Scope: forwards whose launches keep their scratch per call -- the bf16 inference path, which consists of this
repository's kernels only.  The fp32 mode sends its projections to the framework's library GEMMs; three captured
graphs of it replayed side by side did not complete on MI355X / ROCm 7.2 / torch 2.10 (the replays never signalled
their events; one graph at a time is fine), so do not run lanes over a forward that contains library GEMMs.

Shapes are static per ``GraphLanes`` (a serving process keeps one per padded canvas size).  Host constants passed
through the closure (image sizes, canvas) are baked in at capture time like everything else a hipGraph records.
"""
from typing import Callable, List, Sequence

import torch
from torch import Tensor


def _map(struct, fn):
    if isinstance(struct, Tensor):
        return fn(struct)
    if isinstance(struct, (list, tuple)):
        return type(struct)(_map(s, fn) for s in struct)
    raise TypeError("GraphLanes: inputs / outputs must be tensors or (nested) lists / tuples of tensors")


def _zip_apply(dst, src, fn):
    if isinstance(dst, Tensor):
        if not isinstance(src, Tensor) or dst.shape != src.shape or dst.dtype != src.dtype:
            raise ValueError("GraphLanes: input does not match the lane's static tensor "
                             f"({getattr(src, 'shape', None)} / {getattr(src, 'dtype', None)} vs {dst.shape} / {dst.dtype})")
        fn(dst, src)
        return
    if not isinstance(src, (list, tuple)) or len(src) != len(dst):
        raise ValueError("GraphLanes: input structure does not match the example the lanes were captured with")
    for d, s in zip(dst, src):
        _zip_apply(d, s, fn)


class Lane:
    """One captured copy of the forward: ``inputs`` (static, written by ``load``), ``outputs`` (static, overwritten by
    every ``launch``), its ``stream`` and the ``done`` event of its last launch."""

    def __init__(self, fn: Callable, example, capture_kw):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphLanes needs a HIP device (the hot path has no CPU fallback)")
        self.stream = torch.cuda.Stream()
        self.done = torch.cuda.Event()
        self.inputs = _map(example, lambda t: t.clone())
        self._fn = fn
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(2):                  # allocator and lazily built operand caches settle outside the capture
                fn(*self.inputs)
        self.stream.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            with torch.cuda.graph(self.graph, stream=self.stream, **capture_kw), torch.no_grad():
                self.outputs = fn(*self.inputs)
        self.stream.synchronize()

    def load(self, inputs) -> "Lane":
        """Copy ``inputs`` (same structure, shapes and dtypes as the example) into the lane's static tensors, ordered
        after the current stream's work (their producer) and after the lane's previous launch.  The sources may be
        temporaries: each is marked as in use on the lane's stream (``record_stream``), so the caching allocator does
        not hand its memory to new current-stream work before the asynchronous copy has read it."""
        self.stream.wait_stream(torch.cuda.current_stream())

        def copy(d, s):
            d.copy_(s, non_blocking=True)
            if s.is_cuda:
                s.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            _zip_apply(self.inputs, inputs, copy)
        return self

    def launch(self) -> "Lane":
        with torch.cuda.stream(self.stream):
            self.graph.replay()
            self.done.record(self.stream)
        return self

    def wait(self) -> "Lane":
        """The current stream waits for the lane's last launch (no host synchronisation)."""
        torch.cuda.current_stream().wait_event(self.done)
        return self

    def synchronize(self) -> "Lane":
        self.done.synchronize()
        return self


class GraphLanes:
    """``lanes`` captured copies of ``fn(*example_inputs)`` (a no-grad forward with static shapes), taken in turn."""

    def __init__(self, fn: Callable, example_inputs: Sequence, lanes: int = 3, capture_error_mode: str = "global"):
        if lanes < 1:
            raise ValueError("GraphLanes: at least one lane")
        kw = {} if capture_error_mode == "global" else {"capture_error_mode": capture_error_mode}
        self.lanes: List[Lane] = [Lane(fn, tuple(example_inputs), kw) for _ in range(lanes)]
        self._next = 0

    def __len__(self):
        return len(self.lanes)

    def next_lane(self) -> Lane:
        lane = self.lanes[self._next]
        self._next = (self._next + 1) % len(self.lanes)
        return lane

    def submit(self, inputs) -> Lane:
        """Next lane in turn: load ``inputs``, replay.  The lane's previous outputs are overwritten: the caller has
        consumed them (or waited on ``lane.done``) before submitting ``len(self)`` further batches."""
        return self.next_lane().load(tuple(inputs)).launch()

    def launch_next(self) -> Lane:
        """Replay the next lane on the inputs it already holds."""
        return self.next_lane().launch()

    def synchronize(self):
        for lane in self.lanes:
            lane.stream.synchronize()
