"""Dev tool: time emd_conv1x1_f32 on the GEMM shapes of graph D (for rocprofv3 kernel-trace / --pmc)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
SHAPES = {  # name: (B,H,W,Cin,Cout,res)
    "middle": (32, 32, 32, 728, 728, 0),
    "middle_res": (32, 32, 32, 728, 728, 1),
    "dec512_128_64": (32, 512, 512, 128, 64, 0),
    "dec512_64_64_res": (32, 512, 512, 64, 64, 1),
    "dec256_384_128": (32, 256, 256, 384, 128, 0),
    "enc256_128_128": (32, 256, 256, 128, 128, 0),
    "aspp_reduce": (32, 32, 32, 3640, 256, 0),
    "enc64_256_728": (32, 64, 64, 256, 728, 0),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
prec = int(os.environ.get("GB_PREC", "3")); reps = int(os.environ.get("GB_REPS", "10"))
dev = torch.device("cuda", 0)
for nm in names:
    B, H, W, ci, co, res = SHAPES[nm]
    x = ops.Act(torch.rand(B, H, W, ci, device=dev) * 2)
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, ci, co)) * 0.05).astype(np.float32), False, dev)
    s = torch.ones(co, device=dev); t = torch.zeros(co, device=dev)
    out = ops.Act.empty(B, H, W, co, dev)
    r = ops.Act(torch.rand(B, H, W, co, device=dev)) if res else None
    for _ in range(2):
        ops.conv1x1(x, w, s, t, out, res=r, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv1x1(x, w, s, t, out, res=r, precision=prec)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    M = B * H * W
    fl = 2.0 * M * ci * co * (3 if prec == 3 else 1)
    by = 4.0 * M * (ci + co * (2 if res else 1))
    print(f"{nm:18s} M={M:8d} K={ci:4d} N={co:3d}: {us:8.1f} us  {fl/us/1e6:7.1f} TF/s bf16-MFMA ({fl/us/1e6/2500*100:4.1f}% peak)  {by/us/1e3:7.1f} GB/s algorithmic", flush=True)
