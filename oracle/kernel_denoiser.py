"""Graph K oracle: the "dedicated kernel" denoiser (oracle; test infrastructure only).

Restates misc_py/noise-removal-kernels.py:96-431 in numpy (float32 or float64).
PARITY UNPINNED except for KAT #1: with the reference's initial values
(weights 1/w^2, biases 0; noise-removal-kernels.py:109-112) and depth 1 the filter is a
w x w box mean with REFLECT borders.

Parameterisation (make_layer, noise-removal-kernels.py:107-358): a w x w map is built from
``nsym = (o+1)(o+2)/2`` scalars, o = w//2, one per (x, y) with 0 <= y <= x <= o, created in the
order ``for x in range(o+1): for y in range(x+1)`` under the TF names
``depth-{d}_size-{w}/{w0|b1|w1|...}/var_x-{x}_y-{y}/v``; the scalar is shared by the (up to 8)
D4-symmetric positions (+-x,+-y),(+-y,+-x) around the centre.

Forward (filter_fn, noise-removal-kernels.py:378-399) for one w x w patch P of the
REFLECT-padded image (pad, :99-105):
    f = W0 * P
    for i in 1..depth-1:  f = Wi * ( s_i * sigmoid(f + Bi) )      # s_i: bias-free 1->1 fully_connected
    out = sum(f)
The trainer assembles its output transposed (stack axis=1 then axis=2, :421-424) and undoes it
with ``.T`` at :712; like the apply-side class (apply_kernels+MLPs.py:681) this oracle returns
the image un-transposed.
"""
from __future__ import annotations

import numpy as np


def sym_pairs(width: int):
    o = width // 2
    return [(x, y) for x in range(o + 1) for y in range(x + 1)]


def expand_symmetric(vals, width: int) -> np.ndarray:
    """[nsym] scalars (creation order) -> full [w,w] D4-symmetric map."""
    o = width // 2
    pairs = sym_pairs(width)
    vals = np.asarray(vals)
    assert vals.shape == (len(pairs),), (vals.shape, len(pairs))
    lut = {p: k for k, p in enumerate(pairs)}
    full = np.empty((width, width), dtype=vals.dtype)
    for i in range(width):
        for j in range(width):
            a, b = abs(i - o), abs(j - o)
            full[i, j] = vals[lut[(max(a, b), min(a, b))]]
    return full


def init_params(depth: int, width: int, dtype=np.float32):
    """Reference initial values (noise-removal-kernels.py:109-112): weights 1/w^2, biases 0.
    The fully_connected scalars have no reference initial value we can restate exactly
    (glorot-uniform random); 1.0 is used here."""
    n = len(sym_pairs(width))
    return {
        "depth": depth,
        "width": width,
        "w": [np.full(n, 1.0 / (width * width), dtype=dtype) for _ in range(depth)],
        "b": [np.zeros(n, dtype=dtype) for _ in range(depth)],  # b[0] unused
        "s": [dtype(1.0) for _ in range(depth)],  # s[0] unused
    }


def random_params(depth: int, width: int, seed: int, dtype=np.float32):
    rng = np.random.default_rng(seed)
    n = len(sym_pairs(width))
    return {
        "depth": depth,
        "width": width,
        "w": [(rng.standard_normal(n) * (1.5 / (width * width)) + 1.0 / (width * width)).astype(dtype) for _ in range(depth)],
        "b": [(rng.standard_normal(n) * 0.5).astype(dtype) for _ in range(depth)],
        "s": [dtype(rng.uniform(0.5, 2.0)) for _ in range(depth)],
    }


def full_maps(params, dtype=np.float32):
    """-> (W [depth,w,w], Bm [depth,w,w], s [depth]) full symmetric maps."""
    d, w = params["depth"], params["width"]
    W = np.stack([expand_symmetric(np.asarray(params["w"][i], dtype=dtype), w) for i in range(d)])
    Bm = np.stack([expand_symmetric(np.asarray(params["b"][i], dtype=dtype), w) for i in range(d)])
    s = np.asarray(params["s"], dtype=dtype)
    return W, Bm, s


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def denoise_full(x: np.ndarray, W: np.ndarray, Bm: np.ndarray, s: np.ndarray, dtype=np.float32) -> np.ndarray:
    """x [B,H,W,1] -> [B,H,W,1]; W, Bm are FULL [depth,w,w] maps (need not be symmetric)."""
    assert x.ndim == 4 and x.shape[3] == 1
    depth, width = W.shape[0], W.shape[1]
    p = width // 2
    x = x.astype(dtype, copy=False)
    Bn, H, Wd, _ = x.shape
    xp = np.pad(x[..., 0], ((0, 0), (p, p), (p, p)), mode="reflect") if p else x[..., 0]
    out = np.zeros((Bn, H, Wd), dtype=dtype)
    for i in range(width):
        for j in range(width):
            f = dtype(W[0, i, j]) * xp[:, i: i + H, j: j + Wd]
            for l in range(1, depth):
                f = dtype(W[l, i, j]) * (dtype(s[l]) * sigmoid(f + dtype(Bm[l, i, j])))
            out += f.astype(dtype)
    return out[..., None]


def denoise(x: np.ndarray, params, dtype=np.float32) -> np.ndarray:
    W, Bm, s = full_maps(params, dtype)
    return denoise_full(x, W, Bm, s, dtype)


def denoise_loops(x: np.ndarray, params) -> np.ndarray:
    """Pure-Python per-pixel loop that follows the reference's structure literally
    (slice a w x w patch of the padded image per output pixel, :409-417).  Tiny inputs only."""
    W, Bm, s = full_maps(params, np.float64)
    depth, width = W.shape[0], W.shape[1]
    p = width // 2
    Bn, H, Wd, _ = x.shape
    xp = np.pad(x[..., 0].astype(np.float64), ((0, 0), (p, p), (p, p)), mode="reflect")
    out = np.zeros((Bn, H, Wd))
    for b in range(Bn):
        for r in range(H):
            for c in range(Wd):
                f = W[0] * xp[b, r: r + width, c: c + width]
                for l in range(1, depth):
                    f = W[l] * (s[l] * sigmoid(f + Bm[l]))
                out[b, r, c] = f.sum()
    return out[..., None]
