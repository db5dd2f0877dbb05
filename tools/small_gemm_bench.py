"""Dev tool: the training step's pair-sized GEMM launches in isolation (captured into a hipGraph so that host launch overhead is out):
forward / data-gradient GEMM (emd_conv1x1_f32 on fp32 activations, M = 2048, 728 -> 728) and the weight-gradient GEMM of the same layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import _lib, ops, train_ops as TO
dev = torch.device("cuda", 0)
_lib.load()
for (B, H, K, N) in [(2, 32, 728, 728), (8, 32, 728, 728), (2, 64, 256, 728), (2, 128, 256, 256), (2, 512, 64, 64)]:
    x = ops.Act(torch.randn(B, H, H, K, device=dev))
    dy = ops.Act(torch.randn(B, H, H, N, device=dev))
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, K, N)) * 0.05).astype(np.float32), False, dev)
    one, zero = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    out = ops.Act.empty(B, H, H, N, dev)
    dw = torch.zeros(1, K, N, device=dev)
    xs = ops.to_split32(x)
    fns = {"conv1x1": lambda: ops.conv1x1(x, w, one, zero, out, act=False), "conv1x1+stats": lambda: ops.conv_stats(x, w, one, zero, out),
           "split32 conv1x1": lambda: ops.conv1x1_split32(xs, w, one, zero, out, act=False),
           "split32 +stats": lambda: ops.conv1x1_split32(xs, w, one, zero, out, act=False, stats=True),
           "wgrad": lambda: TO.conv_wgrad(x, dy, dw, [0], [0])}
    for name, fn in fns.items():
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 40
        fl = 2.0 * B * H * H * K * N * 3
        print(f"M={B*H*H:7d} K={K} N={N} {name:14s} {us:7.1f} us  {fl/us/1e6:6.1f} TF/s issued", flush=True)
