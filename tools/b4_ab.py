"""Dev tool: graph D at a small batch (the per-GPU share of a strongly scaled global batch of 32) under engine options that matter there."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from emdenoise import _lib
from emdenoise.graphed import GraphedForward
dev = torch.device("cuda", 0)
B = int(os.environ.get("D_B", "4"))
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev, "bf16x3")
x = torch.rand(B, 512, 512, 1, device=dev)
def timed(fn, n=30):
    for _ in range(3): fn(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(x)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = None
for name, env, knobs in [("base", {}, {}), ("sepgemm", {"EMD_D_SEPGEMM": "1"}, {}), ("base", {}, {}), ("sepgemm", {"EMD_D_SEPGEMM": "1"}, {}),
                         ("split_narrow=0", {}, {"split_narrow": 0}), ("dw_th=4", {}, {"dw_th": 4}), ("sep_nw=8", {}, {"sep_nw": 8})]:
    for k, v in env.items(): os.environ[k] = v
    for k, v in knobs.items(): _lib.knob(k, v)
    y = eng.forward(x).clone()
    if ref is None: ref = y
    te = timed(eng.forward)
    tg = timed(GraphedForward(eng))
    print(f"B={B} {name:16s} eager {te:.3f} ms  hipgraph {tg:.3f} ms  same bits as base: {bool(torch.equal(y, ref))}", flush=True)
    for k in env: os.environ.pop(k)
    for k in knobs: _lib.knob(k, {"split_narrow": 1, "dw_th": 0, "sep_nw": 0}[k])
