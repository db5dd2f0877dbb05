"""Teacher-forced per-step comparison: oracle gradients evaluated at the trainer's own parameters."""
import sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emdenoise
from emdenoise import trainer as TR
from oracle import denoiser_graph as G
from tests.synth_inputs import synthetic_pair
from tests.test_train_gpu import weights, flat, rel_l2, cosine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_num_threads(16)
dev = torch.device("cuda", 0)
w = weights()
tr = TR.DenoiserTrainer(w, dev)
names = list(tr.trainable)
for step in range(3):
    lq, hq = synthetic_pair(2, S, S, seed=100 + step)
    cur = tr.state_dict()
    tr.zero_grad()
    res = []
    for k in range(2):
        _, r = tr.tower(torch.from_numpy(lq[k:k+1]).to(dev), torch.from_numpy(hq[k:k+1]).to(dev), update_moving=(k == 0))
        res.append(r.cpu().numpy())
    g = tr.gradients()
    towers = [G.tower_gradients(lq[k:k + 1], hq[k:k + 1], cur, S, dtype=torch.float64) for k in range(2)]
    ref = {n: towers[0]["grads"][n] + towers[1]["grads"][n] for n in names}
    a, b = flat(g, names), flat(ref, names)
    print(f"step {step}: loss {[float(r[1]) for r in res]} vs {[t['loss'] for t in towers]}  grads rel-l2 {rel_l2(a,b):.3e} cos {cosine(a,b):.5f} |g| {np.linalg.norm(b):.3e}", flush=True)
    big = sorted(((np.linalg.norm(g[n].astype(np.float64)-ref[n]), n, np.linalg.norm(ref[n])) for n in names), reverse=True)[:5]
    for e, n, m in big:
        print(f"    abs err {e:.3e}  |ref| {m:.3e}  {n}")
    tr._unpad_grads()
    from emdenoise import train_ops as TO
    TO.nesterov_step(tr.params, tr.grads, tr.accum, 0.001, 0.9, grad_scale=0.5)
    tr.repack()
