"""Dev tool: graph D's forward pass ([32,512,512,1]) under the values of one dev knob, alternating in one process.  D_KNOB=name D_VALS=0,8"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import emdenoise
from emdenoise import _lib

dev = torch.device("cuda", 0)
_lib.load()
knob = os.environ.get("D_KNOB", "sep_nw")
vals = [int(v) for v in os.environ.get("D_VALS", "0,8").split(",")]
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev, "bf16x3")
x = torch.rand(32, 512, 512, 1, device=dev)
T = {v: [] for v in vals}
for v in vals:
    _lib.knob(knob, v)
    eng.forward(x)
torch.cuda.synchronize()
for _ in range(5):
    for v in vals:
        _lib.knob(knob, v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng.forward(x)
        e1.record()
        torch.cuda.synchronize()
        T[v].append(e0.elapsed_time(e1) / 5)
_lib.knob(knob, 0)
print("  ".join(f"{knob}={v}: {np.median(T[v]):.3f} ms (min {min(T[v]):.3f})" for v in vals))
