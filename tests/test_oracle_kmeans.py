"""Pins oracle/csrc/aoc_oracle.c against the real scipy.cluster.vq (third-party dependency of the
reference, call site AEM:276).  Bit-exact: labels, distances and code books."""
import warnings

import numpy as np
import pytest
from scipy.cluster.vq import kmeans2, vq

from oracle import kmeans as okm


def _data(rng, n, d):
    return (np.maximum(rng.randn(n, d), 0) * 0.3).astype(np.float32)


@pytest.mark.parametrize("n,k,d", [(4000, 16, 100), (1500, 64, 100), (333, 7, 100), (2000, 16, 37), (50, 1, 100)])
def test_vq_bit_exact_vs_scipy(n, k, d):
    rng = np.random.RandomState(n + k)
    x = _data(rng, n, d)
    code = x[rng.permutation(n)[:k]].copy()
    if k > 1:   # near-tie: two code rows one ulp apart, and an exact duplicate (tie -> lowest index)
        code[1] = code[0]
        code[1, 3] = np.nextafter(code[1, 3], np.float32(10))
    if k > 3:
        code[3] = code[2]
    lab_s, dist_s = vq(x, code)
    lab_o, low = okm.vq(x, code)
    assert np.array_equal(lab_s, lab_o)
    if k > 1:   # with one code row OpenBLAS takes its gemv path (different summation order);
        #         labels are trivially 0 there, which is all kmeans2 consumes
        assert np.array_equal(dist_s, np.sqrt(np.maximum(low, 0)))   # scipy returns sqrt(max(low,0))


@pytest.mark.parametrize("n,k,d", [(6000, 16, 100), (3000, 64, 100), (40, 16, 100), (900, 8, 100), (700, 32, 24)])
def test_kmeans2_bit_exact_vs_scipy(n, k, d):
    rng = np.random.RandomState(7 * n + k)
    x = _data(rng, n, d)
    if n == 40:
        x[10:30] = x[5]          # few distinct points -> empty clusters keep their previous centroid
    rows = rng.permutation(n)[:k]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cb_s, lab_s = kmeans2(x, x[rows].copy(), minit='matrix', iter=20)
    cb_o, lab_o, counts = okm.kmeans2_matrix(x, x[rows], 20)
    assert np.array_equal(lab_s, lab_o)
    assert np.array_equal(cb_s, cb_o)
    assert np.array_equal(counts, np.bincount(lab_s, minlength=k))


def test_points_init_is_permutation_prefix():
    """minit='points' on the legacy global RandomState == permutation(n)[:k]."""
    x = _data(np.random.RandomState(0), 500, 100)
    np.random.seed(123)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cb_s, lab_s = kmeans2(x, 16, minit='points', iter=20)
    np.random.seed(123)
    rows = okm.draw_init_rows(500, 16)
    cb_o, lab_o, _ = okm.kmeans2_matrix(x, x[rows], 20)
    assert np.array_equal(lab_s, lab_o) and np.array_equal(cb_s, cb_o)


def test_trace_and_errors():
    x = _data(np.random.RandomState(1), 300, 20)
    cb, lab, cnt, tr = okm.kmeans2_matrix(x, x[:4], 5, trace=True)
    assert tr.shape == (5, 300) and np.array_equal(tr[-1], lab)
    with pytest.raises(ValueError):
        okm.kmeans2_matrix(np.full((3, 4), np.nan, np.float32), np.zeros((1, 4), np.float32))
    with pytest.raises(ValueError):
        okm.kmeans2_matrix(np.zeros((0, 4), np.float32), np.zeros((1, 4), np.float32))
