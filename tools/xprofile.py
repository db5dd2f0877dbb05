"""Dev tool: run graph X (or G with XP_GRAPH=G) forward a few times (for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from tests.synth_inputs import synthetic_lq
B = int(os.environ.get("DP_B", "32")); S = int(os.environ.get("DP_S", "512")); n = int(os.environ.get("DP_N", "3"))
dev = torch.device("cuda", 0)
if os.environ.get("XP_GRAPH", "X") == "G":
    from emdenoise import gan
    eng = gan.GeneratorEngine(gan.synthetic_weights(), dev)
else:
    from emdenoise import xception as X
    eng = X.XceptionEngine(X.synthetic_weights(), dev, "bf16x3")
x = torch.from_numpy(np.concatenate([synthetic_lq(2, S, S, seed=1)] * (B // 2 + 1))[:B]).cuda()
y = eng.forward(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    y = eng.forward(x)
torch.cuda.synchronize()
print(f"forward B={B} S={S}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms/step", flush=True)
