"""Time DenoiserTrainer.train_step at 512x512: python tools/train_bench.py [B] [tower_batch] [steps] [streams] [graph 0|1]"""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emdenoise
from emdenoise import denoiser as D, trainer as TR
from tests.synth_inputs import synthetic_pair

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tb = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
streams = int(sys.argv[4]) if len(sys.argv) > 4 else 1
graph = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
S = 512
dev = torch.device("cuda", 0)
w = D.synthetic_weights(variant="Dprime")
tr = TR.DenoiserTrainer(w, dev)
lq, hq = synthetic_pair(2, S, S, seed=1)
lq = torch.from_numpy(np.tile(lq, (B // 2 + 1, 1, 1, 1))[:B]).to(dev)
hq = torch.from_numpy(np.tile(hq, (B // 2 + 1, 1, 1, 1))[:B]).to(dev)
for _ in range(1):
    r = tr.train_step(lq, hq, tower_batch=tb, streams=streams, graph=graph)
torch.cuda.synchronize()
t = time.time()
for _ in range(steps):
    r = tr.train_step(lq, hq, tower_batch=tb, streams=streams, graph=graph)
torch.cuda.synchronize()
dt = (time.time() - t) / steps
print(f"B={B} tower_batch={tb} streams={streams} graph={graph}: {dt*1e3:.1f} ms/step  {B*S*S/dt/1e6:.2f} MPx/s  loss {r[:,1].cpu().numpy()[:4]}  peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
