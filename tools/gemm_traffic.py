"""Dev tool (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE): the split32 pointwise GEMM alone on shapes that tell an over-fetch of
the A operand from a counter artefact -- N = 128 is ONE column tile per row block (nothing to share), N = 728 is six."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
dev = torch.device("cuda", 0)
SH = [(16, 32, 32, 728, 128), (16, 32, 32, 728, 728), (32, 32, 32, 728, 728), (16, 32, 32, 728, 256), (64, 32, 32, 728, 728)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (B, H, W, ci, co) in SH:
    x = ops.Act(torch.rand(B, H, W, ci, device=dev) * 2)
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, ci, co)) * 0.05).astype(np.float32), False, dev)
    s = torch.ones(co, device=dev); t = torch.zeros(co, device=dev)
    xs = ops.to_split32(x)
    o = ops.Act.empty(B, H, W, co, dev)
    os_ = ops.SplitAct(B, H, W, co, dev)
    for k in range(4):
        flush.zero_()          # 512 MiB written in between: the operand comes from HBM, not from the Infinity Cache
        ops.conv1x1_split32(xs, w, s, t, o)
    for k in range(4):
        flush.zero_()
        ops.conv1x1_split32(xs, w, s, t, os_)
    torch.cuda.synchronize()
    M = B * H * W
    print(f"M={M} K={ci} N={co}: A bytes {M * ((ci + 31) // 32 * 32) * 4 / 1e6:.1f} MB, W bytes {co * ((ci+31)//32*32) * 4 / 1e6:.2f} MB, out fp32 {M * co * 4 / 1e6:.1f} MB, out split32 {M * ((co + 31) // 32 * 32) * 4 / 1e6:.1f} MB", flush=True)
