"""Generates tests/golden/dprime_tower_512.json from the ORACLE (oracle/denoiser_graph.py tower_gradients, float64 PyTorch-CPU
autograd; no GPU involved): ONE training tower of graph D' (misc_py/denoiser-multi-gpu.py:752-782: architecture(phase=True),
capped-MSE loss, tf.gradients) on one 512x512 LQ/HQ pair -- BASELINE configs[3]'s size -- with the package's seeded synthetic
weights.  The 38.5 M-element gradient is too large to commit, so the fixture holds
  * mse, loss, 64 probe pixels of the tower's output;
  * the L2 norm of the gradient of every trainable variable;
  * K = 48 projections of the flat gradient onto seeded Rademacher (+-1) vectors: for an error vector e, mean((v_k . e)^2) is an
    unbiased estimate of |e|^2, so relative L2 error and cosine against the full reference gradient can be estimated to ~10 %;
  * the same figures for the oracle's own float32 run against its float64 run ("what any two implementations show").
    python tests/golden/make_train_golden.py        (a few minutes on 8 cores)
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from emdenoise import denoiser as D  # noqa: E402
from oracle import denoiser_graph as G  # noqa: E402
from tests.synth_inputs import synthetic_pair  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K, PSEED = 48, 20261004


def projections(flat):
    """[K] dot products of a float64 vector with the seeded Rademacher vectors."""
    out = []
    for k in range(K):
        sgn = np.random.default_rng(PSEED + k).integers(0, 2, flat.size, dtype=np.int8)
        out.append(float(flat[sgn == 1].sum() - flat[sgn == 0].sum()))
    return out


def main():
    S, seed = 512, 31
    w = D.synthetic_weights(variant="Dprime")
    lq, hq = synthetic_pair(1, S, S, seed=seed)
    t = time.time()
    ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64)
    print(f"float64 tower: {time.time() - t:.1f} s, mse {ref['mse']:.6f} loss {ref['loss']:.6f}", flush=True)
    names = [n for n in ref["grads"] if np.abs(ref["grads"][n]).max() > 1e-9]     # analytically-zero gradients stay out
    flat = np.concatenate([ref["grads"][n].ravel().astype(np.float64) for n in names])
    r32 = G.tower_gradients(lq, hq, w, S, dtype=torch.float32)
    f32 = np.concatenate([r32["grads"][n].ravel().astype(np.float64) for n in names])
    out = ref["out"].numpy()
    rng = np.random.default_rng(0)
    probes = np.stack([rng.integers(0, S, 64), rng.integers(0, S, 64)], axis=1)
    meta = {"S": S, "seed": seed, "lq_sha256": hashlib.sha256(lq.tobytes()).hexdigest(), "mse": ref["mse"], "loss": ref["loss"],
            "probes": probes.tolist(), "out_values": [float(out[0, r, c, 0]) for r, c in probes], "out_mean": float(out.mean()),
            "names": names, "grad_l2": [float(np.linalg.norm(ref["grads"][n].astype(np.float64))) for n in names],
            "flat_l2": float(np.linalg.norm(flat)), "K": K, "projection_seed": PSEED, "projections": projections(flat),
            "oracle_f32_vs_f64": {"rel_l2": float(np.linalg.norm(f32 - flat) / np.linalg.norm(flat)),
                                  "cosine": float(f32 @ flat / (np.linalg.norm(f32) * np.linalg.norm(flat))),
                                  "loss_f32": r32["loss"]}}
    json.dump(meta, open(os.path.join(HERE, "dprime_tower_512.json"), "w"), indent=0)
    print("dprime_tower_512.json:", len(names), "variables, |g| =", meta["flat_l2"], "oracle f32 vs f64:", meta["oracle_f32_vs_f64"])


if __name__ == "__main__":
    main()
