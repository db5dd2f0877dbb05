"""Input-side host logic: batch sharding across GPUs and the reference's numpy feeding conventions.

Mirrors misc_py/denoiser-multi-gpu.py:783-913 (get_scale, gen_lq, scale0to1, flip_rotate, preprocess,
record_parser, input_fn's round-robin sharding) and small_scans/convert_to_numpy.py:11-21 (.npy stacks of
[N,H,W,1] float32).  Everything here is host-side numpy; the GPU path starts at Denoiser.denoise.

Multi-GPU model: one process per GPU; inference shards WHOLE IMAGES across ranks and needs no
collective -- weights are replicated, results stay on the rank that produced them (or are gathered by the
caller).  The reference instead builds in-graph towers and deals images round-robin (`i % num_shards`,
denoiser-multi-gpu.py:905-909); ``shard_round_robin`` reproduces that order, ``shard_contiguous`` is the
per-process form used here.
"""
from __future__ import annotations

import numpy as np


# ---- sharding ---------------------------------------------------------------------------------------------
def shard_contiguous(n: int, world: int, rank: int):
    """Index range [lo, hi) of the `rank`-th of `world` near-equal contiguous shards of n items."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_round_robin(batch, num_shards: int):
    """The reference's input_fn sharding (denoiser-multi-gpu.py:898-913): image i goes to shard i % num_shards;
    returns the list of per-GPU arrays (`feature_shards`)."""
    if num_shards <= 1:
        return [batch]
    return [batch[i::num_shards] for i in range(num_shards)]


def denoise_sharded(denoiser, lq_batch, rank: int, world: int):
    """Run this rank's contiguous shard of a host batch; returns (lo, hi, hq_shard).  No collective."""
    lo, hi = shard_contiguous(len(lq_batch), world, rank)
    if hi == lo:
        return lo, hi, lq_batch[:0]
    return lo, hi, denoiser.denoise_batch(lq_batch[lo:hi])


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """Step time of a multi-rank job = the slowest rank's (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- the reference's host-side image functions ------------------------------------------------------------
def scale0to1(img):
    """denoiser-multi-gpu.py:817-828."""
    img = np.asarray(img)
    lo, hi = np.min(img), np.max(img)
    if lo == hi:
        return np.full(img.shape, 0.5, np.float32)
    return ((img - lo) / (hi - lo)).astype(np.float32)


def get_scale(rng):
    """denoiser-multi-gpu.py:783-784: 25 + Exp(mean 75)."""
    return 25.0 + rng.exponential(75.0)


def gen_lq(img, scale, rng, img_type=np.float32):
    """denoiser-multi-gpu.py:787-799: Poisson counts at `scale`, min-max rescaled."""
    return scale0to1(rng.poisson(np.asarray(img, np.float64) * scale)).astype(img_type)


def flip_rotate(img, choice: int):
    """denoiser-multi-gpu.py:830-851: the 8 elements of D4, selected by `choice` in 0..7."""
    if choice == 0:
        return img
    if choice in (1, 2, 3):
        return np.rot90(img, choice)
    if choice == 4:
        return np.flip(img, 0)
    if choice == 5:
        return np.flip(img, 1)
    if choice == 6:
        return np.flip(np.rot90(img, 1), 0)
    if choice == 7:
        return np.flip(np.rot90(img, 1), 1)
    raise ValueError("choice must be 0..7")


def preprocess(img, rng):
    """denoiser-multi-gpu.py:853-858: NaN/Inf -> 0.5, random D4 element, min-max."""
    img = np.array(img, dtype=np.float32, copy=True)
    img[np.isnan(img)] = 0.5
    img[np.isinf(img)] = 0.5
    return scale0to1(flip_rotate(img, int(8 * rng.random())))


def record_parser(img, rng):
    """denoiser-multi-gpu.py:861-870: (lq, truth rescaled to the lq mean)."""
    img = preprocess(img, rng)
    lq = gen_lq(img, get_scale(rng), rng)
    return lq, ((np.mean(lq) / np.mean(img)) * img).astype(np.float32)


def load_npy_stack(path):
    """small_scans/convert_to_numpy.py:11-21 layout: float32 [N,H,W,1] (mmap, nothing is unpickled)."""
    a = np.load(path, mmap_mode="r", allow_pickle=False)
    if a.ndim == 3:
        a = a[..., None]
    if a.ndim != 4 or a.shape[3] != 1:
        raise ValueError(f"{path}: expected [N,H,W,1], got {a.shape}")
    return a


def input_fn(stack, batch_size: int, num_shards: int, seed: int = 0, epochs: int = 1):
    """Iterator of (feature_shards, truth_shards) lists of per-GPU [n,H,W,1] arrays, like the reference's
    input_fn (denoiser-multi-gpu.py:878-913), from an in-memory / mmapped [N,H,W,1] stack of HQ images."""
    rng = np.random.default_rng(seed)
    n = len(stack)
    for _ in range(epochs):
        order = rng.permutation(n)
        for k in range(0, n - batch_size + 1, batch_size):
            pairs = [record_parser(np.asarray(stack[i])[..., 0], rng) for i in order[k:k + batch_size]]
            lq = np.stack([p[0] for p in pairs])[..., None]
            hq = np.stack([p[1] for p in pairs])[..., None]
            yield shard_round_robin(lq, num_shards), shard_round_robin(hq, num_shards)
