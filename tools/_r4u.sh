python - <<PY
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from emdenoise import _lib, ops, train_ops as TO
dev = torch.device("cuda", 0); _lib.load()
for (B, H, K, N, sa) in [(2, 512, 4, 64, 1), (2, 512, 4, 128, 2)]:
    x = ops.Act(torch.randn(B, H, H, K, device=dev)); Ho = H // sa
    dy = ops.Act(torch.randn(B, Ho, Ho, N, device=dev)); dw = torch.zeros(1, K, N, device=dev)
    fn = lambda: TO.conv_wgrad(x, dy, dw, [0], [0], sa=sa)
    fn(); fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 40
    by = 4.0 * B * Ho * Ho * N + 4.0 * B * H * H * K / (sa * sa)
    print(f"wgrad K=4 N={N} stride {sa} M={B*Ho*Ho}: {us:.1f} us  {by/us/1e3:.0f} GB/s")
PY
