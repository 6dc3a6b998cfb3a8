"""Capture-time guard for hipGraph replay (round 5).

On this stack (ROCm 7.2 / torch 2.10) a ``hipMemsetAsync`` captured into a graph is NOT reproduced by replay: the node
exists, the replayed graph leaves the memory as it was.  Round 4 found three victims the hard way -- the MSDA backward's
work counter, ATen's multi-block ``reduce_kernel`` (its semaphores are cleared by a memset) and a captured fused AdamW
(CHANGELOG.md, round 4) -- each of which produced a replayed training step that silently differed from the eager one.
``memset_nodes(graph)`` lists the memset nodes of a captured ``torch.cuda.CUDAGraph`` (the graph has to be created with
``keep_graph=True`` so that the hipGraph_t stays alive) and ``assert_replay_safe`` raises when there are any: a fill that
has to happen inside a captured region must be a kernel (``tensor.zero_()`` / ``fill_`` are; ``torch.zeros`` of a fresh
block, ``hipMemsetAsync`` and the semaphore reset of ATen's split reductions are not).
"""
import ctypes
from typing import List

import torch

_HIP_GRAPH_NODE_TYPE_MEMSET = 2          # hipGraphNodeTypeMemset (hip_runtime_api.h)
_hip_rt = None


def _runtime():
    global _hip_rt
    if _hip_rt is None:
        # the HIP runtime torch itself is linked against (already mapped into the process)
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _hip_rt = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hip_rt is None:
            raise RuntimeError("graph_guard: the HIP runtime (libamdhip64.so) is not loadable")
        _hip_rt.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip_rt.hipGraphGetNodes.restype = ctypes.c_int
        _hip_rt.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _hip_rt.hipGraphNodeGetType.restype = ctypes.c_int
    return _hip_rt


def new_graph() -> "torch.cuda.CUDAGraph":
    """A ``torch.cuda.CUDAGraph`` whose hipGraph_t survives instantiation, so that its nodes can be inspected."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:       # older torch: no handle to inspect
        return torch.cuda.CUDAGraph()


def node_types(graph: "torch.cuda.CUDAGraph") -> List[int]:
    """hipGraphNodeType of every top-level node of a captured graph (``new_graph()``); [] when torch exposes no handle."""
    raw = getattr(graph, "raw_cuda_graph", None)
    if raw is None:
        return []
    try:
        handle = raw()
    except RuntimeError:     # created without keep_graph
        return []
    rt = _runtime()
    n = ctypes.c_size_t(0)
    if rt.hipGraphGetNodes(ctypes.c_void_p(handle), None, ctypes.byref(n)) != 0 or n.value == 0:
        return []
    nodes = (ctypes.c_void_p * n.value)()
    if rt.hipGraphGetNodes(ctypes.c_void_p(handle), nodes, ctypes.byref(n)) != 0:
        return []
    out = []
    for i in range(n.value):
        t = ctypes.c_int(-1)
        rt.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        out.append(t.value)
    return out


def memset_nodes(graph: "torch.cuda.CUDAGraph") -> int:
    return sum(1 for t in node_types(graph) if t == _HIP_GRAPH_NODE_TYPE_MEMSET)


def assert_replay_safe(graph: "torch.cuda.CUDAGraph", what: str = "captured region") -> int:
    """Raises if the captured graph holds memset nodes (not reproduced by replay on this stack); returns the number of
    nodes inspected (0: this torch exposes no graph handle -- nothing could be checked)."""
    types = node_types(graph)
    bad = sum(1 for t in types if t == _HIP_GRAPH_NODE_TYPE_MEMSET)
    if bad:
        raise RuntimeError(f"{what}: {bad} memset node(s) among the {len(types)} captured nodes -- hipGraph replay does not "
                           "reproduce them on this stack (salience_detr_amd/graph_guard.py); make the fill a kernel")
    return len(types)
