import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.set_num_threads(16)
from emdenoise import xception as X
from oracle import xception_graph as XG
from tests.synth_inputs import synthetic_lq
B, S = int(os.environ.get("XB", "2")), int(os.environ.get("XS", "128"))
w = X.synthetic_weights()
x = synthetic_lq(B, S, S, seed=7)
tr_o, tr_p = [], []
ref = XG.architecture(x, w, S, dtype=torch.float64, trace=tr_o).numpy()
got = X.XceptionEngine(w, torch.device("cuda", 0)).forward(torch.from_numpy(x).cuda(), trace=tr_p).cpu().numpy()
print("layers traced", len(tr_o), len(tr_p))
worst = 0
for i, (a, b) in enumerate(zip(tr_o, tr_p)):
    a = a.numpy()
    r = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30)
    if r > 3 * worst or i % 12 == 0 or r > 1e-3:
        print(f"  #{i:3d} {str(a.shape):22s} rel {r:.2e}")
    worst = max(worst, r)
print("final rel", np.linalg.norm(got - ref) / np.linalg.norm(ref))
