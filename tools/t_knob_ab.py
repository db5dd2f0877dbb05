"""Dev tool: the D' training step (8 pairs of 512 x 512 per GPU, batched per-image towers, hipGraph) under dev-knob / environment
settings, each on a fresh trainer (a captured graph bakes the launch parameters in), alternating, one process.
T_SPECS="wgrad_msplit=1;wgrad_tile=64;wgrad_msplit=1,wgrad_tile=64;env:EMD_T_GROUPS=2" (the base setting always runs first and last)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from emdenoise import _lib, denoiser as D, trainer as TR
from tests.synth_inputs import synthetic_pair
dev = torch.device("cuda", 0)
B, S = 8, 512
w = D.synthetic_weights(variant="Dprime")
lq, hq = synthetic_pair(2, S, S, seed=1)
lq = torch.from_numpy(np.tile(lq, (B // 2 + 1, 1, 1, 1))[:B]).to(dev)
hq = torch.from_numpy(np.tile(hq, (B // 2 + 1, 1, 1, 1))[:B]).to(dev)
specs = ["base"] + [s for s in os.environ.get("T_SPECS", "").split(";") if s] + ["base"]
DEFAULTS = {}
def apply(spec, on):
    if spec == "base": return
    for kv in spec.split(","):
        if kv.startswith("env:"):
            k, v = kv[4:].split("=")
            if on: os.environ[k] = v
            else: os.environ.pop(k, None)
        else:
            k, v = kv.split("=")
            _lib.knob(k, int(v) if on else 0)
ref = None
for spec in specs:
    apply(spec, True)
    tr = TR.DenoiserTrainer(w, dev)
    for _ in range(2):
        r = tr.train_step(lq, hq, tower_batch=1, streams=8, graph=True, batched=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(4):
            r = tr.train_step(lq, hq, tower_batch=1, streams=8, graph=True, batched=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) / 4 * 1e3)
    g = tr.grads.clone()
    if ref is None: ref = g
    rel = float((g - ref).norm() / ref.norm())
    print(f"{spec:40s} {np.median(ts):7.2f} ms/step (min {min(ts):.2f})  loss {float(r[0,1]):.6f}  params-grad rel diff vs first {rel:.1e}", flush=True)
    apply(spec, False)
    del tr
    torch.cuda.empty_cache()
