"""Dev tool: wall-clock split of one training step (batched towers, hipGraph): graph replay / optimizer / re-pack."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from emdenoise import trainer as TR
from tests.synth_inputs import synthetic_pair
dev = torch.device("cuda", 0)
w = emdenoise.synthetic_weights(variant="Dprime")
lq, hq = synthetic_pair(8, 512, 512, seed=3)
x, t = torch.from_numpy(lq).to(dev), torch.from_numpy(hq).to(dev)
tr = TR.DenoiserTrainer(w, dev)
for _ in range(2): tr.train_step(x, t, tower_batch=1, graph=True, batched=True)
torch.cuda.synchronize()
def timed(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
key = list(tr._graphs.keys())[0]
g = tr._graphs[key][0]
print(f"whole train_step      {timed(lambda: tr.train_step(x, t, tower_batch=1, graph=True, batched=True)):7.2f} ms")
print(f"graph replay          {timed(lambda: g.replay()):7.2f} ms")
from emdenoise import train_ops as TO
print(f"nesterov_step         {timed(lambda: TO.nesterov_step(tr.params, tr.grads, tr.accum, 1e-9, 0.9, grad_scale=1.0 / 8)):7.2f} ms")
print(f"repack                {timed(lambda: tr.repack()):7.2f} ms")
