"""Dev tool: single-image latency of graph D, eager launches vs a captured hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from tests.synth_inputs import synthetic_lq
dev = torch.device("cuda", 0)
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev)
for B in (1, 4):
    x = torch.from_numpy(synthetic_lq(B, 512, 512, seed=1)).to(dev)
    for _ in range(3): y = eng.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): y = eng.forward(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    g = torch.cuda.CUDAGraph()
    xs = x.clone()
    with torch.cuda.graph(g):
        ys = eng.forward(xs)
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(ys, y)
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
    print(f"B={B}: eager {eager*1e3:.2f} ms, hipGraph replay {graph*1e3:.2f} ms, identical output: {ok}", flush=True)
