"""Dev tool: run one K-kernel variant (EMD_K_VARIANT) for correctness + timing under rocprofv3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emdenoise
from oracle import kernel_denoiser as KO

B = int(os.environ.get("KT_B", "32"))
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
D = int(os.environ.get("KT_DEPTH", "2"))
params = KO.random_params(D, 3, seed=3)
W, Bm, s = KO.full_maps(params)
p = emdenoise.KernelParams(W, Bm, s)
pd = torch.from_numpy(p.packed()).to(dev)
for shape in [(2, 72, 512), (1, 40, 1024), (3, 33, 64)]:
    x = rng.random(shape + (1,)).astype(np.float32)
    ref = KO.denoise(x, params, np.float64)
    y = emdenoise.kernel_denoise(torch.from_numpy(x).to(dev), pd, 3, D, True).cpu().numpy()
    rel = np.linalg.norm(y - ref) / np.linalg.norm(ref)
    assert rel < 2e-6 or os.environ.get('KT_NOCHECK'), (shape, rel)
x = torch.rand(B, 512, 512, 1, device=dev)
y = torch.empty_like(x)
for _ in range(20):
    emdenoise.kernel_denoise(x, pd, 3, D, True, out=y)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        emdenoise.kernel_denoise(x, pd, 3, D, True, out=y)
g.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"variant {os.environ.get('EMD_K_VARIANT','0')}: ok, graph-replay {dt*1e6:.2f} us/launch -> {8*B*512*512/dt/1e9:.0f} GB/s")
