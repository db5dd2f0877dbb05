"""Dev tool: A/B of the register-staged pointwise GEMM (emd_conv1x1_f32) against the LDS-DMA split32 GEMM
(emd_conv1x1_split32_f32), interleaved rounds in one process; also the depthwise kernel with fp32 / split32 output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
SHAPES = {  # name: (B,H,W,Cin,Cout,res)
    "middle": (32, 32, 32, 728, 728, 0),
    "middle_res": (32, 32, 32, 728, 728, 1),
    "enc64_256_728": (32, 64, 64, 256, 728, 0),
    "enc64_728_728": (32, 64, 64, 728, 728, 0),
    "dec128_384_256": (32, 128, 128, 384, 256, 0),
    "dec128_256_256": (32, 128, 128, 256, 256, 0),
    "dec256_384_128": (32, 256, 256, 384, 128, 0),
    "enc256_128_128": (32, 256, 256, 128, 128, 0),
    "x_exit_1024": (32, 16, 16, 728, 1024, 0),
    "x_exit_1536": (32, 8, 8, 1536, 1536, 0),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
reps = int(os.environ.get("GB_REPS", "10")); rounds = int(os.environ.get("GB_ROUNDS", "5"))
dev = torch.device("cuda", 0)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for nm in names:
    B, H, W, ci, co, res = SHAPES[nm]
    x = ops.Act(torch.rand(B, H, W, ci, device=dev) * 2)
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, ci, co)) * 0.05).astype(np.float32), False, dev)
    dw = torch.rand(9, ci, device=dev)
    s = torch.ones(co, device=dev); t = torch.zeros(co, device=dev)
    o1 = ops.Act.empty(B, H, W, co, dev); o2 = ops.Act.empty(B, H, W, co, dev)
    r = ops.Act(torch.rand(B, H, W, co, device=dev)) if res else None
    xs = ops.to_split32(x)
    d1 = ops.Act.empty(B, H, W, ci, dev); d2 = ops.SplitAct(B, H, W, ci, dev)
    fns = {"v2": lambda: ops.conv1x1(x, w, s, t, o1, res=r), "v3": lambda: ops.conv1x1_split32(xs, w, s, t, o2, res=r),
           "cvt": lambda: ops.to_split32(x, xs), "dw": lambda: ops.dw3x3(x, dw, d1), "dws": lambda: ops.dw3x3_split32(x, dw, d2)}
    for f in fns.values(): f(); f()
    torch.cuda.synchronize()
    same = bool(torch.equal(o1.buf, o2.buf))
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items(): T[k].append(timed(f))
    M = B * H * W
    fl = 6.0 * M * ci * co
    med = {k: float(np.median(v)) for k, v in T.items()}
    print(f"{nm:16s} M={M:8d} K={ci:4d} N={co:4d} same={same}: v2 {med['v2']:7.1f} us ({fl/med['v2']/1e6/25:4.1f}%)  v3 {med['v3']:7.1f} us "
          f"({fl/med['v3']/1e6/25:4.1f}% of 2.5 PF issued; min {min(T['v3']):.1f})  cvt {med['cvt']:6.1f}  dw {med['dw']:6.1f}  dw_split {med['dws']:6.1f}", flush=True)
