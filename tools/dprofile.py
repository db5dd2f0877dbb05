"""Dev tool: run graph D forward a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from tests.synth_inputs import synthetic_lq
B = int(os.environ.get("DP_B", "32")); S = int(os.environ.get("DP_S", "512")); prec = os.environ.get("DP_PREC", "bf16x3")
n = int(os.environ.get("DP_N", "3"))
eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), torch.device("cuda", 0), prec, fuse_sep=os.environ.get("DP_FUSE", "1") == "1")
x = torch.from_numpy(np.concatenate([synthetic_lq(2, S, S, seed=1)] * (B // 2 + 1))[:B]).cuda()
y = eng.forward(x); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    y = eng.forward(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"D forward B={B} S={S} {prec}: {dt*1e3:.2f} ms/step -> {B*S*S/1e6/dt:.1f} MPx/s; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
