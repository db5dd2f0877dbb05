import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from emdenoise import train_ops as TO
from oracle import gan_graph as GG
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
n = 100003
p = rng.uniform(-0.3, 0.3, n); g = rng.standard_normal(n) * 10 ** rng.uniform(-7, -1, n)
m = np.zeros(n); v = np.zeros(n)
P, G, M, V = (torch.from_numpy(a.astype(np.float32)).to(dev) for a in (p, g, m, v))
for step in (1, 2, 3):
    gn2 = TO.sumsq(G, scale=0.5)
    TO.adam_step(P, G, M, V, step, 1e-4, beta1=0.5, grad_scale=0.5, gnorm_sq=gn2, clip_norm=15.0)
    own, gn = GG.clip_by_global_norm({"a": 0.5 * G.cpu().numpy().astype(np.float64)}, 15.0)
    newp, mm, vv = GG.adam_step({"a": p}, own, {"a": m}, {"a": v}, step, 1e-4)
    got = P.cpu().numpy().astype(np.float64)
    print(step, "gn", gn, float(gn2.sqrt()), "update rel", np.linalg.norm((got - p) - (newp["a"] - p)) / np.linalg.norm(newp["a"] - p),
          "m rel", np.linalg.norm(M.cpu().numpy() - mm["a"]) / np.linalg.norm(mm["a"]))
    p, m, v = newp["a"], mm["a"], vv["a"]
    P.copy_(torch.from_numpy(p.astype(np.float32)))
