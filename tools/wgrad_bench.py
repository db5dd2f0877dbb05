"""Dev tool: time emd_conv_wgrad_f32 on the layer shapes of a D' training step (tower of 8 and of 1 image)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import emdenoise
from emdenoise import ops, train_ops as TO
dev = torch.device("cuda", 0)
shapes = [  # (B,H,W,K,N,taps,rate)
    (8, 32, 32, 728, 728, 1, 1), (1, 32, 32, 728, 728, 1, 1), (8, 32, 32, 728, 728, 9, 6), (1, 32, 32, 728, 728, 9, 6),
    (8, 512, 512, 64, 64, 1, 1), (1, 512, 512, 64, 64, 1, 1), (8, 256, 256, 128, 128, 1, 1), (8, 128, 128, 384, 256, 1, 1),
    (8, 32, 32, 3640, 256, 1, 1), (8, 64, 64, 256, 728, 1, 1),
]
for (B, H, W, K, N, taps, rate) in shapes:
    a = ops.Act(torch.randn(B, H, W, K, device=dev))
    dy = ops.Act(torch.randn(B, H, W, N, device=dev))
    dw = torch.zeros(taps, K, N, device=dev)
    tdy, tdx = (TO.conv_taps(H, W, 1, rate) if taps == 9 else ([0], [0]))
    for _ in range(2):
        TO.conv_wgrad(a, dy, dw, tdy, tdx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        TO.conv_wgrad(a, dy, dw, tdy, tdx)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * H * W * K * N * taps
    by = 4.0 * B * H * W * (K + N)
    print(f"B{B} {H}x{W} K{K} N{N} taps{taps}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e3:7.1f} GB/s(A+dY once)")
