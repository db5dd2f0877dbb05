"""Generates the committed golden vectors of graph D from the ORACLE (float64), no GPU involved:
  tests/golden/d_graph_64.npz    input + full output for a [2,64,64,1] batch
  tests/golden/d_graph_512.json  64 probe pixels per image, mean and SHA-256 of the float32 output for a
                                 [2,512,512,1] batch at BASELINE's crop size (full tensors are too large)
The weights are the package's seeded synthetic set (emdenoise.synthetic_weights()).
    python tests/golden/make_d_golden.py
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

import emdenoise  # noqa: E402
from oracle import denoiser_graph as G  # noqa: E402
from tests.synth_inputs import synthetic_lq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    w = emdenoise.synthetic_weights()
    x = synthetic_lq(2, 64, 64, seed=64)
    y = G.architecture(x, w, 64, dtype=torch.float64).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "d_graph_64.npz"), x=x, y=y)
    print("d_graph_64.npz", y.shape, float(y.mean()))

    B, seed = 2, 512
    x = synthetic_lq(B, 512, 512, seed=seed)
    t = time.time()
    y = G.architecture(x, w, 512, dtype=torch.float64).numpy()
    print(f"512x512 float64 oracle: {time.time() - t:.1f} s")
    rng = np.random.default_rng(0)
    probes = np.stack([rng.integers(0, B, 128), rng.integers(0, 512, 128), rng.integers(0, 512, 128)], axis=1)
    probes[:8] = [[0, 0, 0], [0, 0, 511], [0, 511, 0], [0, 511, 511], [1, 0, 0], [1, 255, 256], [1, 511, 511], [1, 0, 1]]
    meta = {
        "B": B, "seed": seed, "x_sha256": hashlib.sha256(x.tobytes()).hexdigest(),
        "y_f32_sha256": hashlib.sha256(y.astype(np.float32).tobytes()).hexdigest(),
        "mean": float(y.mean()), "std": float(y.std()),
        "probes": probes.tolist(), "values": [float(y[b, r, c, 0]) for b, r, c in probes],
    }
    json.dump(meta, open(os.path.join(HERE, "d_graph_512.json"), "w"), indent=0)
    print("d_graph_512.json mean", meta["mean"], "std", meta["std"])


if __name__ == "__main__":
    main()
