"""Generates tests/golden/x_graph_512.json from the ORACLE (oracle/xception_graph.py, float64; no GPU involved): graph X
(misc_py/modified_Xception.py:194-654) at BASELINE's crop size on a [2,512,512,1] batch with the package's seeded synthetic
weights (calibrated at 512 px, data/synth_bn_X_seed1234.npz):
  * 128 probe pixels of the output, its mean / std and the SHA-256 of the float32 output;
  * for every traced tensor (101: each separable block, conv block, transposed conv) its L2 norm, mean and the values at 4
    fixed positions, so that the GPU test can check the whole chain -- decoder included -- at this size layer by layer;
  * the oracle's own float32-vs-float64 relative L2 at the output and per traced tensor ("noise32": what any float32-class
    implementation shows).
    python tests/golden/make_x_golden.py         (about 10 minutes on 8 cores)
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

from emdenoise import xception as X  # noqa: E402
from oracle import xception_graph as XG  # noqa: E402
from tests.synth_inputs import synthetic_lq  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def layer_positions(shape):
    B, H, W, C = shape
    return [(0, 0, 0, 0), (B - 1, H - 1, W - 1, C - 1), (0, H // 2, W // 3, C // 2), (B - 1, H // 3, W // 2, C // 3)]


class Stats(list):
    """A trace sink that keeps statistics, not tensors; `f32` = the float32 run's tensors, compared on the fly."""

    def __init__(self, f32):
        super().__init__()
        self.f32 = f32

    def append(self, t):
        pos = layer_positions(tuple(t.shape))
        l2 = float(torch.linalg.vector_norm(t))
        a = self.f32[len(self)]
        noise = float(torch.linalg.vector_norm(a.double() - t)) / max(l2, 1e-30)
        self.f32[len(self)] = None
        super().append({"shape": list(t.shape), "l2": l2, "mean": float(t.mean()), "values": [float(t[p]) for p in pos],
                        "noise32": noise, "zeros": float((t == 0).double().mean())})


def main():
    B, S, seed = 2, 512, 512
    w = X.synthetic_weights()
    x = synthetic_lq(B, S, S, seed=seed)
    t = time.time()
    t32 = []
    y32 = XG.architecture(x, w, S, dtype=torch.float32, trace=t32).numpy()
    print(f"float32 oracle: {time.time() - t:.1f} s", flush=True)
    t = time.time()
    st = Stats(t32)
    y = XG.architecture(x, w, S, dtype=torch.float64, trace=st).numpy()
    print(f"float64 oracle: {time.time() - t:.1f} s, output mean {y.mean():.4f} std {y.std():.4f} zeros {(y == 0).mean():.3f} ones {(y == 1).mean():.3f}", flush=True)
    noise32 = float(np.linalg.norm(y32.astype(np.float64) - y) / np.linalg.norm(y))
    print(f"noise32 at the output {noise32:.3e}; per layer max {max(s['noise32'] for s in st):.3e}", flush=True)
    for i, s_ in enumerate(st):
        print(i, s_["shape"], f"noise32 {s_['noise32']:.2e} zeros {s_['zeros']:.2f} mean {s_['mean']:.3g}")
    rng = np.random.default_rng(0)
    probes = np.stack([rng.integers(0, B, 128), rng.integers(0, S, 128), rng.integers(0, S, 128)], axis=1)
    probes[:8] = [[0, 0, 0], [0, 0, S - 1], [0, S - 1, 0], [0, S - 1, S - 1], [1, 0, 0], [1, S // 2 - 1, S // 2], [1, S - 1, S - 1], [1, 0, 1]]
    meta = {"B": B, "S": S, "seed": seed, "x_sha256": hashlib.sha256(x.tobytes()).hexdigest(),
            "y_f32_sha256": hashlib.sha256(y.astype(np.float32).tobytes()).hexdigest(),
            "mean": float(y.mean()), "std": float(y.std()), "l2": float(np.linalg.norm(y)), "noise32": noise32,
            "probes": probes.tolist(), "values": [float(y[b, r, c, 0]) for b, r, c in probes], "layers": list(st)}
    json.dump(meta, open(os.path.join(HERE, "x_graph_512.json"), "w"), indent=0)
    print("x_graph_512.json written:", len(st), "layers")


if __name__ == "__main__":
    main()
