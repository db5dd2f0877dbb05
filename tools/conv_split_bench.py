"""Dev tool: A/B of the register-staged implicit-GEMM convolutions (emd_conv3x3_f32 / emd_deconv3x3s2_f32) against the
LDS-DMA split32 forms, interleaved rounds in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
SHAPES = {  # name: (kind, B, H, W, Cin, Cout)
    "D_deconv2to1": ("deconv", 32, 128, 128, 256, 256),
    "D_deconv1to0": ("deconv", 32, 256, 256, 128, 128),
    "X_c3_256x128": ("conv", 32, 256, 256, 128, 128),
    "X_c3_128x192": ("conv", 32, 128, 128, 192, 192),
    "X_c3_64x256": ("conv", 32, 64, 64, 256, 256),
    "X_c3_32x384": ("conv", 32, 32, 32, 384, 384),
    "X_dc_128to256": ("deconv", 32, 128, 128, 192, 128),
    "X_dc_64to128": ("deconv", 32, 64, 64, 256, 192),
    "Dp_aspp_r6": ("conv6", 32, 32, 32, 728, 728),
    "X_c3_512x64": ("conv", 32, 512, 512, 64, 64),
    "X_dc_256to512": ("deconv", 32, 256, 256, 128, 64),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
reps = int(os.environ.get("GB_REPS", "5")); rounds = int(os.environ.get("GB_ROUNDS", "3"))
dev = torch.device("cuda", 0)
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for nm in names:
    kind, B, H, W, ci, co = SHAPES[nm]
    x = ops.Act(torch.rand(B, H, W, ci, device=dev) * 2)
    xs = ops.to_split32(x)
    s = torch.ones(co, device=dev); t = torch.zeros(co, device=dev)
    rng = np.random.default_rng(0)
    if kind == "deconv":
        w = ops.pack_deconv((rng.standard_normal((3, 3, co, ci)) * 0.03).astype(np.float32), dev)
        o1 = ops.Act.empty(B, 2 * H, 2 * W, co, dev); o2 = ops.Act.empty(B, 2 * H, 2 * W, co, dev); o3 = ops.SplitAct(B, 2 * H, 2 * W, co, dev)
        fns = {"v2": lambda: ops.deconv3x3s2(x, w, s, t, o1), "split->f32": lambda: ops.deconv3x3s2_split32(xs, w, s, t, o2),
               "split->split": lambda: ops.deconv3x3s2_split32(xs, w, s, t, o3)}
        flops = 6.0 * B * H * W * ci * co * 9
    else:
        rate = 6 if kind == "conv6" else 1
        w = ops.PackedWeights((rng.standard_normal((9, ci, co)) * 0.03).astype(np.float32), False, dev)
        o1 = ops.Act.empty(B, H, W, co, dev); o2 = ops.Act.empty(B, H, W, co, dev); o3 = ops.SplitAct(B, H, W, co, dev)
        fns = {"v2": lambda: ops.conv3x3(x, w, s, t, o1, rate=rate), "split->f32": lambda: ops.conv3x3_split32(xs, w, s, t, o2, rate=rate),
               "split->split": lambda: ops.conv3x3_split32(xs, w, s, t, o3, rate=rate)}
        flops = 6.0 * B * H * W * ci * co * 9
    fns["cvt"] = lambda: ops.to_split32(x, xs)
    for f in fns.values(): f(); f()
    torch.cuda.synchronize()
    same = bool(torch.equal(o1.buf, o2.buf))
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items(): T[k].append(timed(f))
    med = {k: float(np.median(v)) for k, v in T.items()}
    print(f"{nm:16s} same={same}: " + "  ".join(f"{k} {med[k]:8.1f} us" + (f" ({flops/med[k]/1e6/25:4.1f}%)" if k != "cvt" else "") for k in fns), flush=True)
