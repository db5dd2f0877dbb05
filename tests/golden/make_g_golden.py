"""Generates tests/golden/g_graph_64.npz from the ORACLE (float64), no GPU involved: input (1/64-sampled, -1
elsewhere) + full output of the in-filling generator (misc_py/gan-infilling-100.py:133-374) for a [2,64,64,1] batch,
with the package's seeded synthetic weights (emdenoise.gan.synthetic_weights()).
    python tests/golden/make_g_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import emdenoise  # noqa: E402,F401
from emdenoise import gan as GN  # noqa: E402
from oracle import gan_graph as GG  # noqa: E402
from tests.synth_inputs import synthetic_lq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    w = GN.synthetic_weights()
    x = GN.gen_lq(2.0 * synthetic_lq(2, 64, 64, seed=64)[..., 0] - 1.0)[..., None]
    y = GG.generator(x, w, 64, dtype=torch.float64).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g_graph_64.npz"), x=x, y=y)
    print("g_graph_64.npz", y.shape, float(y.mean()), float(y.std()))


if __name__ == "__main__":
    main()
