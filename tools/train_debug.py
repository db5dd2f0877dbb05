"""Per-variable gradient comparison: DenoiserTrainer.tower (GPU) vs the oracle's autograd (CPU float64)."""
import sys, time
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emdenoise
from emdenoise import denoiser as D, trainer as TR
from oracle import denoiser_graph as G
from tests.synth_inputs import synthetic_pair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
torch.set_num_threads(16)
w = D.synthetic_weights(variant="Dprime")
lq, hq = synthetic_pair(B, S, S, seed=3)
t = time.time()
ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64)
print("oracle f64 %.1fs mse %.6f loss %.6f" % (time.time() - t, ref["mse"], ref["loss"]), flush=True)
ref32 = G.tower_gradients(lq, hq, w, S, dtype=torch.float32)
print("oracle f32 done", flush=True)
dev = torch.device("cuda", 0)
tr = TR.DenoiserTrainer(w, dev)
tr.zero_grad()
print("trainer built", flush=True)
out, res = tr.tower(torch.from_numpy(lq).to(dev), torch.from_numpy(hq).to(dev))
torch.cuda.synchronize()
print("tower done", flush=True)
res = res.cpu().numpy()
print("gpu mse %.6f loss %.6f" % (res[0], res[1]))
rl = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))
print("out rel-l2 gpu %.2e  oracle32 %.2e" % (rl(out.cpu().numpy(), ref["out"].numpy()), rl(ref32["out"].numpy(), ref["out"].numpy())))
g = tr.gradients()
tot_g = np.concatenate([g[n].reshape(-1) for n in g]); tot_r = np.concatenate([ref["grads"][n].reshape(-1) for n in g])
tot_32 = np.concatenate([ref32["grads"][n].reshape(-1) for n in g])
print("ALL grads rel-l2 gpu %.2e oracle32 %.2e" % (rl(tot_g, tot_r), rl(tot_32, tot_r)))
worst = []
zero_abs = 0.0
for n in g:
    r = ref["grads"][n]
    if np.abs(r).max() < 1e-9:      # analytically zero (bias / beta in front of a batch norm)
        zero_abs = max(zero_abs, float(np.abs(g[n]).max()))
        continue
    worst.append((rl(g[n], r), rl(ref32["grads"][n], r), n, float(np.abs(r).max())))
print("analytically-zero gradients: max |gpu| = %.2e" % zero_abs)
order = list(g)
for e, e32, n, m in sorted(worst, reverse=True)[:12]:
    print("%.2e  (oracle32 %.2e)  max|g| %.2e  %s" % (e, e32, m, n))
print("-- in graph order (every 12th)")
for e, e32, n, m in worst[::12]:
    print("%.2e  (oracle32 %.2e)  max|g| %.2e  %s" % (e, e32, m, n))
st = tr.state_dict()
mw = max((rl(st[n], v), n) for n, v in ref["moving"].items())
print("moving stats worst rel-l2", mw)
