"""Generates tests/golden/s_graph_160.npz from the ORACLE (float64), no GPU involved: input + full output of the small
separable autoencoder (misc_py/apply_autoencoders.py:91-187) for a [2,160,160,1] batch (per-image batch statistics, as
the reference's one-crop sess.run gives them), encoding_features = 16, with the package's seeded synthetic weights
(emdenoise.autoencoder.synthetic_weights()).
    python tests/golden/make_s_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import emdenoise  # noqa: E402,F401
from emdenoise import autoencoder as AE  # noqa: E402
from oracle import autoencoder_graph as AG  # noqa: E402
from tests.synth_inputs import synthetic_lq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    w = AE.synthetic_weights(16)
    x = synthetic_lq(2, 160, 160, seed=160)
    x = x / x.mean(axis=(1, 2, 3), keepdims=True)       # the class divides by the image mean (:353)
    y = AG.architecture(x, w, 16, dtype=torch.float64).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "s_graph_160.npz"), x=x.astype(np.float32), y=y)
    print("s_graph_160.npz", y.shape, float(y.mean()), float(y.std()))


if __name__ == "__main__":
    main()
