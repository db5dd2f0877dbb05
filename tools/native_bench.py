"""Dev tool: graph D [32,512,512,1] through the native executor (emd_graph_run) and through the Python engine, interleaved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from emdenoise.graph_exec import NativeGraph
from tests.synth_inputs import synthetic_lq
dev = torch.device("cuda", 0)
w = emdenoise.synthetic_weights()
eng = emdenoise.DenoiserEngine(w, dev, "bf16x3")
nat = NativeGraph(w, dev)
x = torch.from_numpy(np.concatenate([synthetic_lq(2, 512, 512, seed=1)] * 16)).to(dev)
a, b = eng.forward(x), nat.forward(x)
torch.cuda.synchronize()
print("bit-identical:", bool(torch.equal(a, b)))
def timed(fn, n=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for two in (False, True, False, True):
    nat.set_two_streams(two)
    print(f"python engine {timed(lambda: eng.forward(x)):6.2f} ms   native executor ({'two streams' if two else 'one stream'}) {timed(lambda: nat.forward(x)):6.2f} ms", flush=True)
