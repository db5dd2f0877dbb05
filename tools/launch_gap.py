"""Dev tool: per-launch time of a trivially small kernel inside a replayed hipGraph (the floor under tools/small_gemm_bench.py's figures),
and of the small GEMMs with 1 / 4 independent launches in flight (4 streams captured into one graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import _lib, ops, train_ops as TO
dev = torch.device("cuda", 0)
_lib.load()
def timed(fn, n=20, rep=2):
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (n * rep)
a, b = ops.Act(torch.zeros(1, 8, 8, 4, device=dev)), ops.Act(torch.zeros(1, 8, 8, 4, device=dev))
print(f"tiny kernel (axpy on 256 floats): {timed(lambda: TO.axpy(a, b)):.1f} us per launch")
B, H, K, N = 2, 32, 728, 728
xs = [ops.Act(torch.randn(B, H, H, K, device=dev)) for _ in range(4)]
dys = [ops.Act(torch.randn(B, H, H, N, device=dev)) for _ in range(4)]
dws = [torch.zeros(1, K, N, device=dev) for _ in range(4)]
outs = [ops.Act.empty(B, H, H, N, dev) for _ in range(4)]
w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, K, N)) * 0.05).astype(np.float32), False, dev)
one, zero = torch.ones(N, device=dev), torch.zeros(N, device=dev)
print(f"wgrad M=2048, one stream: {timed(lambda: TO.conv_wgrad(xs[0], dys[0], dws[0], [0], [0])):.1f} us per launch")
print(f"conv1x1 M=2048, one stream: {timed(lambda: ops.conv1x1(xs[0], w, one, zero, outs[0], act=False)):.1f} us per launch")
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
def four(fn):
    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                for _ in range(5):
                    fn(i)
        for s in streams:
            cur.wait_stream(s)
    return run
t = timed(four(lambda i: TO.conv_wgrad(xs[i], dys[i], dws[i], [0], [0])), n=4)
print(f"wgrad M=2048, 4 streams x 5 launches: {t / 20 * 1:.1f} us per launch of 20 ({t:.0f} us per group of 20)")
t = timed(four(lambda i: ops.conv1x1(xs[i], w, one, zero, outs[i], act=False)), n=4)
print(f"conv1x1 M=2048, 4 streams x 5 launches: {t / 20:.1f} us per launch of 20 ({t:.0f} us per group of 20)")
