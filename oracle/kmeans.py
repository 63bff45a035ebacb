"""ctypes front-end of oracle/csrc/aoc_oracle.c (TEST INFRASTRUCTURE ONLY).

Restates scipy.cluster.vq.kmeans2 as called at
/root/reference/AOC-Net/adaptive_embedding_for_matching.py:276 (``minit='points',
iter=20``): see the header of aoc_oracle.c for the algorithm and its arithmetic order.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libaoc_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/csrc/aoc_oracle.c -> oracle/libaoc_oracle.so (gcc, ~1 s)."""
    src = os.path.join(_HERE, "csrc", "aoc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-f", os.path.join(_HERE, "Makefile")])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.aoc_oracle_vq.argtypes = [vp, vp, ci, ci, ci, vp, vp]
        L.aoc_oracle_vq.restype = None
        L.aoc_oracle_update_means.argtypes = [vp, vp, ci, ci, ci, vp, vp]
        L.aoc_oracle_update_means.restype = None
        L.aoc_oracle_kmeans2.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp]
        L.aoc_oracle_kmeans2.restype = ci
        L.aoc_oracle_dense_match_min.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
        L.aoc_oracle_dense_match_min.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def vq(obs, code):
    """scipy ``_vq.vq``: returns (labels int32[n], squared low distance float32[n])."""
    obs = np.ascontiguousarray(obs, np.float32)
    code = np.ascontiguousarray(code, np.float32)
    n, d = obs.shape
    k = code.shape[0]
    labels = np.empty(n, np.int32)
    low = np.empty(n, np.float32)
    lib().aoc_oracle_vq(_p(obs), _p(code), n, k, d, _p(labels), _p(low))
    return labels, low


def kmeans2_matrix(obs, init, iters=20, trace=False):
    """``kmeans2(obs, init, minit='matrix', iter=iters)`` -> (code_book, labels, counts[, trace]).

    ``counts`` are the member counts of the last update (0 = cluster kept its old centroid).
    """
    obs = np.ascontiguousarray(obs, np.float32)
    code = np.array(init, np.float32, order="C", copy=True)
    n, d = obs.shape
    k = code.shape[0]
    if n < 1 or k < 1:
        raise ValueError("Empty input is not supported.")  # vq.py:793-794
    if not np.isfinite(obs).all():
        raise ValueError("array must not contain infs or NaNs")  # check_finite=True
    labels = np.empty(n, np.int32)
    counts = np.empty(k, np.int32)
    tr = np.empty((iters, n), np.int32) if trace else None
    rc = lib().aoc_oracle_kmeans2(_p(obs), _p(code), n, k, d, int(iters), _p(labels), _p(counts), _p(tr))
    if rc != 0:
        raise ValueError("kmeans2 oracle: invalid arguments")
    return (code, labels, counts, tr) if trace else (code, labels, counts)


def draw_init_rows(n, k, rng=None):
    """Row indices that ``minit='points'`` picks: ``rng.choice(n, k, replace=False)``
    on the legacy global RandomState == ``permutation(n)[:k]`` (vq.py:499-522)."""
    rng = np.random if rng is None else rng
    return np.asarray(rng.permutation(n)[:k], np.int64)


def dense_match_min_scalar(q, r, wrong_bits, n_obj):
    """Single-thread scalar port of AEM:61-89 (used only as a cores=1 CPU baseline)."""
    q = np.ascontiguousarray(q, np.float32)
    r = np.ascontiguousarray(r, np.float32)
    wrong_bits = np.ascontiguousarray(wrong_bits, np.uint32)
    m, d = q.shape
    n = r.shape[0]
    out = np.empty((m, n_obj), np.float32)
    lib().aoc_oracle_dense_match_min(_p(q), _p(r), _p(wrong_bits), m, n, d, n_obj, _p(out))
    return out
