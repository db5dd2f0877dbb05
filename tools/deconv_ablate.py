"""Dev tool: what a phase of the patch-resident transposed conv costs (knob sep_ablate; results wrong on purpose)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

dev = torch.device("cuda", 0)
_lib.load()
for nm, (B, H, W, ci, co) in {"deconv1to0": (32, 256, 256, 128, 128), "deconv2to1": (32, 128, 128, 256, 256)}.items():
    x = ops.to_split32(ops.Act(torch.rand(B, H, W, ci, device=dev)))
    w = ops.pack_deconv((np.random.default_rng(0).standard_normal((3, 3, co, ci)) * 0.03).astype(np.float32), dev)
    s, t = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    out = ops.Act.empty(B, 2 * H, 2 * W, co, dev)
    for label, bits in (("full (product build)", 0), ("full (ablation build)", 1), ("-mfma", 2), ("-stores", 4), ("-mfma-stores", 6), ("-patch DMA", 8), ("-weight DMA", 16), ("-all DMA", 24),
                        ("DMA + barriers only", 6), ("nothing but barriers", 30)):
        _lib.knob("sep_ablate", bits)
        for _ in range(2):
            ops.deconv3x3s2_fused(x, w, s, t, out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.deconv3x3s2_fused(x, w, s, t, out)
        e1.record()
        torch.cuda.synchronize()
        print(f"{nm:12s} {label:22s}: {e0.elapsed_time(e1) * 200:8.1f} us", flush=True)
    _lib.knob("sep_ablate", 0)
