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
import os
from typing import List

import torch

_HIP_GRAPH_NODE_TYPE_MEMSET = 2          # hipGraphNodeTypeMemset (hip_runtime_api.h)
_HIP_GRAPH_NODE_TYPE_GRAPH = 4           # hipGraphNodeTypeGraph: a child graph
_hip_rt = None


def _mapped_runtime_path():
    """Path of the libamdhip64 ALREADY mapped into this process (the one torch's graphs belong to): handles of one
    runtime instance must not be handed to a second copy loaded by an unversioned name."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "libamdhip64.so" in os.path.basename(path):
                    return path
    except OSError:
        pass
    return None


def _runtime():
    global _hip_rt
    if _hip_rt is None:
        path = _mapped_runtime_path()
        if path is None:
            raise RuntimeError("graph_guard: no libamdhip64 is mapped into this process (is torch's HIP runtime initialised?)")
        _hip_rt = ctypes.CDLL(path)          # same inode as the mapped copy: the loader returns that instance
        _hip_rt.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
        _hip_rt.hipGraphGetNodes.restype = ctypes.c_int
        _hip_rt.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _hip_rt.hipGraphNodeGetType.restype = ctypes.c_int
        _hip_rt.hipGraphChildGraphNodeGetGraph.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        _hip_rt.hipGraphChildGraphNodeGetGraph.restype = ctypes.c_int
    return _hip_rt


def new_graph() -> "torch.cuda.CUDAGraph":
    """A ``torch.cuda.CUDAGraph`` whose hipGraph_t survives instantiation, so that its nodes can be inspected."""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:       # older torch: no handle to inspect
        return torch.cuda.CUDAGraph()


def _graph_node_types(rt, handle, out: List[int], depth: int = 0) -> None:
    n = ctypes.c_size_t(0)
    status = rt.hipGraphGetNodes(ctypes.c_void_p(handle), None, ctypes.byref(n))
    if status != 0:
        raise RuntimeError(f"graph_guard: hipGraphGetNodes failed with status {status}")
    if n.value == 0:
        return
    nodes = (ctypes.c_void_p * n.value)()
    status = rt.hipGraphGetNodes(ctypes.c_void_p(handle), nodes, ctypes.byref(n))
    if status != 0:
        raise RuntimeError(f"graph_guard: hipGraphGetNodes failed with status {status}")
    for i in range(n.value):
        t = ctypes.c_int(-1)
        status = rt.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        if status != 0:
            raise RuntimeError(f"graph_guard: hipGraphNodeGetType failed with status {status}")
        out.append(t.value)
        if t.value == _HIP_GRAPH_NODE_TYPE_GRAPH and depth < 8:      # a child graph's memsets count too
            child = ctypes.c_void_p()
            if rt.hipGraphChildGraphNodeGetGraph(ctypes.c_void_p(nodes[i]), ctypes.byref(child)) == 0 and child.value:
                _graph_node_types(rt, child.value, out, depth + 1)


def node_types(graph: "torch.cuda.CUDAGraph") -> List[int]:
    """hipGraphNodeType of every node of a captured graph (``new_graph()``), child graphs included; [] when torch exposes
    no handle.  A failing runtime call raises (an empty list must mean "no handle", never "could not look")."""
    raw = getattr(graph, "raw_cuda_graph", None)
    if raw is None:
        return []
    try:
        handle = raw()
    except RuntimeError:     # created without keep_graph
        return []
    out: List[int] = []
    _graph_node_types(_runtime(), handle, out)
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
