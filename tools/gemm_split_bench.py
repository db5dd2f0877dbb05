"""Dev tool: A/B of the register-staged pointwise GEMM (emd_conv1x1_f32) against the LDS-DMA split32 GEMM
(emd_conv1x1_split32_f32), interleaved rounds in one process; also the depthwise kernel with fp32 / split32 output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops, _lib
import ctypes as C
lib = _lib.load()
VARIANTS = [int(v) for v in os.environ.get('GB_VARIANTS', '3,4').split(',')]
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
    def v3(v):
        def f():
            lib.emd_debug_split_variant(v)
            ops.conv1x1_split32(xs, w, s, t, o2, res=r)
        return f
    fns = {"v2": lambda: ops.conv1x1(x, w, s, t, o1, res=r)}
    for v in VARIANTS: fns[f"v3.{v}"] = v3(v)
    if os.environ.get("GB_AUX", "1") == "1":
        fns.update({"cvt": lambda: ops.to_split32(x, xs), "dw": lambda: ops.dw3x3(x, dw, d1), "dws": lambda: ops.dw3x3_split32(x, dw, d2)})
    same = []
    fns["v2"]()
    for v in VARIANTS:
        o2.buf.zero_(); fns[f"v3.{v}"](); torch.cuda.synchronize()
        eq = bool(torch.equal(o1.buf, o2.buf))
        same.append(eq if eq else f"rel {float((o1.buf - o2.buf).norm() / o1.buf.norm()):.1e}")
    for f in fns.values(): f(); f()
    torch.cuda.synchronize()
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items(): T[k].append(timed(f))
    M = B * H * W
    fl = 6.0 * M * ci * co
    med = {k: float(np.median(v)) for k, v in T.items()}
    print(f"{nm:16s} M={M:8d} K={ci:4d} N={co:4d} same={same}: " + "  ".join(
        f"{k} {med[k]:7.1f} us" + (f" ({fl/med[k]/1e6/25:4.1f}%)" if k[0] == "v" else "") for k in fns), flush=True)
    if os.environ.get("GB_STAMPS"):
        for v in VARIANTS:
            bm = 128 if v == 2 else 256
            nblk = -(-M // bm) * -(-co // 128)
            if v == 4: nblk = min(nblk, 256)
            st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
            lib.emd_debug_split_stamps(C.c_void_p(st.data_ptr())); fns[f"v3.{v}"](); torch.cuda.synchronize()
            lib.emd_debug_split_stamps(C.c_void_p(0))
            a = st.view(nblk, 8).cpu().numpy().astype(np.float64)
            ph = np.stack([a[:, 1] - a[:, 0], a[:, 2] - a[:, 1], a[:, 3] - a[:, 2], a[:, 4] - a[:, 3]], 1)
            print(f"   stamps v3.{v}: blocks {nblk}; cycles median/max: first-tile {np.median(ph[:,0]):.0f}/{ph[:,0].max():.0f}  k-loop {np.median(ph[:,1]):.0f}/{ph[:,1].max():.0f} "
                  f"({np.median(ph[:,1])/(-(-ci//32)):.0f} per step)  acc->LDS {np.median(ph[:,2]):.0f}/{ph[:,2].max():.0f}  stores {np.median(ph[:,3]):.0f}/{ph[:,3].max():.0f}  "
                  f"whole {np.median(a[:,4]-a[:,0]):.0f}", flush=True)
            if v == 4: print(f"      persistent: first tile's K loop {np.median(a[:,3]-a[:,1]):.0f} cycles ({np.median(a[:,3]-a[:,1])/(-(-ci//32)):.0f} per step); rest of the tile loop {np.median(a[:,2]-a[:,3]):.0f}; flush {np.median(a[:,4]-a[:,2]):.0f}")
            ai = st.view(nblk, 8).cpu().numpy()
            xcc = (ai[:, 5] >> 32) & 15; hw = ai[:, 5] & 0xffffffff
            cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
            ncu = len(np.unique(cu)); per = np.bincount(np.unique(cu, return_inverse=True)[1])
            span_rt = (ai[:, 7].max() - ai[:, 6].min()) / 100.0   # us (100 MHz)
            clk = np.median((a[:, 4] - a[:, 0]) / np.maximum(ai[:, 7] - ai[:, 6], 1) * 100.0)
            spans = [a[xcc == x, 4].max() - a[xcc == x, 0].min() for x in np.unique(xcc)]
            print(f"      CUs used {ncu}, blocks per CU min/max {per.min()}/{per.max()}; first start -> last end {span_rt:.1f} us; in-kernel clock {clk:.0f} MHz; "
                  f"per-XCD span cycles {np.min(spans):.0f}..{np.max(spans):.0f}", flush=True)
