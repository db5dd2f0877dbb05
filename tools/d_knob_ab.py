"""Dev tool: graph D's forward pass ([32,512,512,1]) under the values of dev knobs, alternating in one process (box-to-box differences
are 2 %; inside one process the medians repeat to 0.02 ms).  D_SPECS="sep_nw:0,8;epi_width:0,4" (the first value of a knob is restored)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import emdenoise
from emdenoise import _lib

dev = torch.device("cuda", 0)
_lib.load()
B = int(os.environ.get("D_B", "32"))
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev, "bf16x3")
x = torch.rand(B, 512, 512, 1, device=dev)
for spec in os.environ.get("D_SPECS", "sep_nw:0,8").split(";"):
    knob, vs = spec.split(":")
    vals = [int(v) for v in vs.split(",")]
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
    _lib.knob(knob, vals[0])
    print("  ".join(f"{knob}={v}: {np.median(T[v]):.3f} ms (min {min(T[v]):.3f})" for v in vals), flush=True)
