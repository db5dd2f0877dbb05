"""CPU tests of the graph-K oracle (oracle/kernel_denoiser.py, oracle/k_oracle.c).

Pins (parity is otherwise unpinned): KAT #1 -- at the reference's initial values
and depth 1 the filter is a w x w box mean with REFLECT borders
(misc_py/noise-removal-kernels.py:109-112) -- and agreement of three independent statements of
the graph: vectorised numpy, a literal per-pixel Python loop, and plain C.
"""
import ctypes

import numpy as np
import pytest

from oracle import kernel_denoiser as K


def run_c(lib, x, W, Bm, s, nthreads=2):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    W = np.ascontiguousarray(W, dtype=np.float32)
    Bm = np.ascontiguousarray(Bm, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float32)
    rc = lib.k_oracle_f32(x.ctypes.data, y.ctypes.data, x.shape[0], x.shape[1], x.shape[2], W.shape[1], W.shape[0],
                          W.ctypes.data, Bm.ctypes.data, s.ctypes.data, nthreads)
    assert rc == 0
    return y


def test_sym_pairs_and_expand():
    assert K.sym_pairs(3) == [(0, 0), (1, 0), (1, 1)]          # centre, edge, corner
    assert len(K.sym_pairs(5)) == 6 and len(K.sym_pairs(15)) == 36
    m = K.expand_symmetric(np.array([1.0, 2.0, 3.0]), 3)
    np.testing.assert_array_equal(m, [[3, 2, 3], [2, 1, 2], [3, 2, 3]])
    m5 = K.expand_symmetric(np.arange(6.0), 5)
    assert np.array_equal(m5, m5.T) and np.array_equal(m5, m5[::-1]) and np.array_equal(m5, m5[:, ::-1])
    assert m5[2, 2] == 0 and m5[2, 3] == 1 and m5[3, 3] == 2 and m5[2, 4] == 3 and m5[3, 4] == 4 and m5[4, 4] == 5


@pytest.mark.parametrize("width", [3, 5])
def test_kat_box_mean(width, k_oracle_lib):
    rng = np.random.default_rng(0)
    x = rng.random((2, 9, 11, 1))
    p = width // 2
    xp = np.pad(x[..., 0], ((0, 0), (p, p), (p, p)), mode="reflect")
    expect = np.zeros((2, 9, 11))
    for i in range(width):
        for j in range(width):
            expect += xp[:, i:i + 9, j:j + 11]
    expect /= width * width
    params = K.init_params(1, width, np.float64)
    got = K.denoise(x, params, np.float64)[..., 0]
    np.testing.assert_allclose(got, expect, rtol=1e-13)
    W, Bm, s = K.full_maps(K.init_params(1, width))
    got_c = run_c(k_oracle_lib, x, W, Bm, s)[..., 0]
    np.testing.assert_allclose(got_c, expect, rtol=2e-6)


@pytest.mark.parametrize("depth,width", [(1, 3), (2, 3), (3, 3), (2, 5), (5, 7)])
def test_three_statements_agree(depth, width, k_oracle_lib):
    rng = np.random.default_rng(depth * 10 + width)
    x = rng.random((2, 8, 9, 1)) * 2.0
    params = K.random_params(depth, width, seed=depth + width, dtype=np.float64)
    a = K.denoise(x, params, np.float64)
    b = K.denoise_loops(x, params)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)
    W, Bm, s = K.full_maps(params)
    c = run_c(k_oracle_lib, x, W, Bm, s)
    f32 = K.denoise(x.astype(np.float32), params, np.float32)
    np.testing.assert_allclose(c, a, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(f32, a, rtol=2e-5, atol=2e-6)


def test_c_rejects_bad_arguments(k_oracle_lib):
    x = np.zeros((1, 4, 4), np.float32)
    y = np.zeros_like(x)
    w = np.zeros(9, np.float32)
    s = np.zeros(1, np.float32)
    f = k_oracle_lib.k_oracle_f32
    assert f(x.ctypes.data, y.ctypes.data, 1, 4, 4, 4, 1, w.ctypes.data, w.ctypes.data, s.ctypes.data, 1) == -1  # even width
    assert f(None, y.ctypes.data, 1, 4, 4, 3, 1, w.ctypes.data, w.ctypes.data, s.ctypes.data, 1) == -1
    assert f(x.ctypes.data, y.ctypes.data, 1, 1, 4, 3, 1, w.ctypes.data, w.ctypes.data, s.ctypes.data, 1) == -1  # pad >= dim
