"""Dev tool: a whole forward pass as two half batches on two streams vs one batch (graph D / G / X-less)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import emdenoise
from tests.synth_inputs import synthetic_lq
dev = torch.device("cuda", 0)
which = os.environ.get("HB_GRAPH", "D")
if which == "G":
    from emdenoise import gan
    eng = gan.GeneratorEngine(gan.synthetic_weights(), dev)
else:
    eng = emdenoise.DenoiserEngine(emdenoise.synthetic_weights(), dev, "bf16x3")
B = 32
x = torch.from_numpy(synthetic_lq(B, 512, 512, seed=1)).to(dev)
s = [torch.cuda.Stream(), torch.cuda.Stream()]
main = torch.cuda.current_stream()
def whole():
    return eng.forward(x)
def halves(stagger):
    outs = []
    for h in range(2):
        s[h].wait_stream(main)
        with torch.cuda.stream(s[h]):
            outs.append(eng.forward(x[h * 16:(h + 1) * 16]))
    for h in range(2):
        main.wait_stream(s[h])
    return outs
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for inner in (True, False):
    eng.two_streams = inner
    print(f"{which} inner two_streams={inner}: whole {timeit(whole):.2f} ms   two half batches on two streams {timeit(lambda: halves(0)):.2f} ms", flush=True)
