"""ctypes/numpy front-end of the plain-C MSDA oracle (TEST INFRASTRUCTURE).

Mirrors the reference op signatures (models/bricks/ops/cuda/ms_deform_attn_cuda.cu:12-18,
75-82) on numpy arrays.  See oracle/msda_oracle.c for the arithmetic.
"""
import ctypes
import os

import numpy as np

from . import build_oracle

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = build_oracle.LIB
        if not os.path.exists(path):
            path = build_oracle.build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_msda_num_threads.restype = ctypes.c_int
    return _LIB


def num_threads() -> int:
    return int(_lib().oracle_msda_num_threads())


def _prep(value, shapes, lsi, loc, aw):
    dt = value.dtype
    assert dt in (np.float32, np.float64)
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    aw = np.ascontiguousarray(aw, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    lsi = np.ascontiguousarray(lsi, dtype=np.int64)
    B, Nv, M, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    assert aw.shape == (B, Nq, M, L, P) and shapes.shape == (L, 2) and lsi.shape == (L,)
    return value, shapes, lsi, loc, aw, (B, Nv, M, D, L, Nq, P)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def msda_forward(value, shapes, lsi, loc, aw):
    value, shapes, lsi, loc, aw, dims = _prep(value, shapes, lsi, loc, aw)
    B, Nv, M, D, L, Nq, P = dims
    out = np.empty((B, Nq, M * D), dtype=value.dtype)
    fn = getattr(_lib(), "oracle_msda_forward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), *[ctypes.c_int(d) for d in dims], _p(out))
    return out


def msda_backward(value, shapes, lsi, loc, aw, grad_out):
    value, shapes, lsi, loc, aw, dims = _prep(value, shapes, lsi, loc, aw)
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    gv = np.zeros_like(value)
    gl = np.zeros_like(loc)
    ga = np.zeros_like(aw)
    fn = getattr(_lib(), "oracle_msda_backward_" + ("f32" if value.dtype == np.float32 else "f64"))
    fn(_p(value), _p(shapes), _p(lsi), _p(loc), _p(aw), _p(grad_out),
       *[ctypes.c_int(d) for d in dims], _p(gv), _p(gl), _p(ga))
    return gv, gl, ga
