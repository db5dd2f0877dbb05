"""Dev tool: time emd_sep3x3_fused_f32 on graph D's fused separable layers (EMD_SEP_TPW knob; SB_STAMPS=1 prints the in-kernel phase split)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
SHAPES = {"deconv0_a": (512, 128, 64, 0), "deconv0_b": (512, 64, 64, 1), "deconv1_a": (256, 384, 128, 0), "cnn1": (256, 128, 128, 0),
          "deconv1_b": (256, 128, 128, 1)}
dev = torch.device("cuda", 0)
B = int(os.environ.get("SB_B", "32"))
for nm in (sys.argv[1].split(",") if len(sys.argv) > 1 else SHAPES):
    S, ci, co, res = SHAPES[nm]
    x = ops.Act(torch.rand(B, S, S, ci, device=dev)); w = torch.rand(9, ci, device=dev)
    pw = ops.PackedWeights(np.random.default_rng(0).standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
    s1, t1 = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    out = ops.Act.empty(B, S, S, co, dev)
    r = ops.Act(torch.rand(B, S, S, co, device=dev)) if res else None
    for _ in range(2): ops.sep_fused(x, w, pw, s1, t1, out, res=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.sep_fused(x, w, pw, s1, t1, out, res=r)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    by = 4.0 * B * S * S * (ci + co * (2 if res else 1))
    if os.environ.get("SB_STAMPS"):
        import ctypes
        from emdenoise import _lib
        lib = _lib.load()
        nwg = B * (S // 8) * (S // 16)
        st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        lib.emd_debug_sep_stamps.argtypes = [ctypes.c_void_p]; lib.emd_debug_sep_stamps.restype = None
        lib.emd_debug_sep_stamps(ctypes.c_void_p(st.data_ptr()))
        ops.sep_fused(x, w, pw, s1, t1, out, res=r); torch.cuda.synchronize()
        lib.emd_debug_sep_stamps(ctypes.c_void_p(0))
        v = st.view(-1, 8).double(); v = v[v.sum(1) > 0]
        m = v.mean(0)
        names = ["lds-write+bar1", "issue loads", "depthwise", "wk loads+bar2", "mfma", "bar3", "epilogue", "-"]
        print("   stamps (100 MHz ticks per workgroup, %d wgs): total %.0f | " % (v.shape[0], m.sum().item()) + "  ".join(f"{n} {100*a/m.sum().item():.0f}%" for n, a in zip(names, m.tolist()) if a > 0), flush=True)
    print(f"tpw={os.environ.get('EMD_SEP_TPW','-')} {nm:10s}: {us:8.1f} us  {by/us/1e3:7.1f} GB/s algorithmic ({by/us/1e3/8000*100:4.1f}% of 8 TB/s)", flush=True)
