"""Dev tool: graph D at [32,512,512,1] with the 1/16-resolution flow run in 2 (default) / 4 / 8 parts (EMD_D_PARTS), one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
dev = torch.device("cuda", 0)
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev, "bf16x3")
x = torch.rand(32, 512, 512, 1, device=dev)
ref = None
for rnd in range(2):
    for parts in ("2", "4", "8"):
        os.environ["EMD_D_PARTS"] = parts
        y = eng.forward(x).clone()
        if ref is None: ref = y
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): eng.forward(x)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        print(f"parts={parts}: {np.median(ts):.3f} ms (min {min(ts):.3f})  same bits: {bool(torch.equal(y, ref))}", flush=True)
