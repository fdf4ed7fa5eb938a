"""TEST INFRASTRUCTURE ONLY -- Python face of the C search oracle (oracle/search_ref.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  ``build()`` compiles the C file with gcc into ``oracle/_build/``.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmmr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "search_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_SO)
        f32p, i64p, f64p = (ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int64),
                            ctypes.POINTER(ctypes.c_double))
        lib.mmr_ref_dot64.restype = ctypes.c_double
        lib.mmr_ref_dot64.argtypes = [f32p, f32p, ctypes.c_int]
        lib.mmr_ref_cosine_topk.restype = None
        lib.mmr_ref_cosine_topk.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_float, i64p, f32p, f64p]
        lib.mmr_ref_similarity.restype = None
        lib.mmr_ref_similarity.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_float, f32p]
        lib.mmr_ref_topk_merge.restype = None
        lib.mmr_ref_topk_merge.argtypes = [i64p, f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, i64p, f32p, f64p]
        lib.mmr_ref_l2norm_rows.restype = None
        lib.mmr_ref_l2norm_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int]
        _lib = lib
    return _lib


def _f32(x) -> np.ndarray:
    """Widen any float tensor/array to contiguous fp32 (exact for bf16/fp16/fp32)."""
    if isinstance(x, torch.Tensor):
        x = x.detach().to("cpu").to(torch.float32).numpy()
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def cosine_topk(q, gallery, k: int, scale: float = 1.0):
    """Exact (-score,+index)-ordered top-k.  Returns (idx int64[Q,k], score f32[Q,k], dot f64[Q,k])."""
    lib = _load()
    q, g = _f32(q), _f32(gallery)
    if q.ndim == 1:
        q = q[None, :]
    Q, E = q.shape
    N = g.shape[0]
    assert g.shape[1] == E
    idx = np.empty((Q, k), np.int64)
    score = np.empty((Q, k), np.float32)
    s64 = np.empty((Q, k), np.float64)
    lib.mmr_ref_cosine_topk(_p(q, ctypes.c_float), _p(g, ctypes.c_float), Q, N, E, k, float(scale),
                            _p(idx, ctypes.c_int64), _p(score, ctypes.c_float), _p(s64, ctypes.c_double))
    return idx, score, s64


def similarity(q, gallery, scale: float = 1.0) -> np.ndarray:
    """Materialised ``scale * q @ gallery.T`` -> f32[Q,N] with the oracle's fixed-order fp64 dot."""
    lib = _load()
    q, g = _f32(q), _f32(gallery)
    if q.ndim == 1:
        q = q[None, :]
    Q, E = q.shape
    N = g.shape[0]
    out = np.empty((Q, N), np.float32)
    lib.mmr_ref_similarity(_p(q, ctypes.c_float), _p(g, ctypes.c_float), Q, N, E, float(scale),
                           _p(out, ctypes.c_float))
    return out


def topk_merge(idx_parts: np.ndarray, s64_parts: np.ndarray, scale: float = 1.0):
    """Merge [parts,Q,k] per-shard lists (global ids, fp64 dots)."""
    lib = _load()
    idx_parts = np.ascontiguousarray(idx_parts, np.int64)
    s64_parts = np.ascontiguousarray(s64_parts, np.float64)
    parts, Q, k = idx_parts.shape
    idx = np.empty((Q, k), np.int64)
    score = np.empty((Q, k), np.float32)
    s64 = np.empty((Q, k), np.float64)
    lib.mmr_ref_topk_merge(_p(idx_parts, ctypes.c_int64), _p(s64_parts, ctypes.c_double), parts, Q, k,
                           float(scale), _p(idx, ctypes.c_int64), _p(score, ctypes.c_float),
                           _p(s64, ctypes.c_double))
    return idx, score, s64


def l2norm_rows(x) -> np.ndarray:
    lib = _load()
    x = _f32(x).copy()
    lib.mmr_ref_l2norm_rows(_p(x, ctypes.c_float), x.shape[0], x.shape[1])
    return x


def reference_expression_topk(features: torch.Tensor, ref: torch.Tensor, k: int, scale: float = 100.0):
    """The reference's own torch expression, verbatim in meaning:
    ``similarity = 100. * features @ ref_feature.t()`` (code/search_image.py:107) followed by
    ``output.topk(k, 1, True, True)`` (code/utils.py:17) with queries as rows.
    fp32 on CPU; summation order and tie order are whatever torch/MKL pick, so this is used to
    cross-check the exact oracle on tie-free data and as the timed CPU baseline -- never as the
    bit-exact checker."""
    sim = scale * features.float() @ ref.float().t()       # [N,Q]
    vals, idx = sim.t().topk(k, 1, True, True)             # [Q,k]
    return idx, vals
