"""Seeded synthetic low-quality micrograph crops shared by the tests, the golden-vector scripts and
smoke() (SURVEY.md 8d "Synthetic inputs"): smooth field of random Gaussians -> Poisson counts with
scale = 25 + Exp(75) (misc_py/denoiser-multi-gpu.py:783-799) -> min-max to [0,1]."""
import numpy as np


def synthetic_lq(B, H, W, seed=1234):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    out = np.empty((B, H, W, 1), np.float32)
    for b in range(B):
        hq = np.zeros((H, W))
        for _ in range(8):
            cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(H / 64 + 1, H / 8 + 2)
            hq += rng.uniform(0.2, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
        hq = (hq - hq.min()) / max(hq.max() - hq.min(), 1e-9)
        lq = rng.poisson(hq * (25.0 + rng.exponential(75.0))).astype(np.float64)
        out[b, :, :, 0] = ((lq - lq.min()) / max(lq.max() - lq.min(), 1e-9)).astype(np.float32)
    return out


def synthetic_pair(B, H, W, seed=1234):
    """(lq, truth) training pairs, as gen_lq builds them (misc_py/denoiser-multi-gpu.py:785-812): lq as above, truth =
    the clean image rescaled by mean(lq)/mean(clean)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    lq_out = np.empty((B, H, W, 1), np.float32)
    hq_out = np.empty((B, H, W, 1), np.float32)
    for b in range(B):
        hq = np.zeros((H, W))
        for _ in range(8):
            cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(H / 64 + 1, H / 8 + 2)
            hq += rng.uniform(0.2, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
        hq = (hq - hq.min()) / max(hq.max() - hq.min(), 1e-9)
        lq = rng.poisson(hq * (25.0 + rng.exponential(75.0))).astype(np.float64)
        lq = (lq - lq.min()) / max(lq.max() - lq.min(), 1e-9)
        lq_out[b, :, :, 0] = lq.astype(np.float32)
        hq_out[b, :, :, 0] = (hq * lq.mean() / max(hq.mean(), 1e-9)).astype(np.float32)
    return lq_out, hq_out
