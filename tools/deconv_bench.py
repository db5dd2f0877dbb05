"""Dev tool: the transposed convs of graphs D and X through emd_deconv3x3s2_fused_split32_f32 -- dev knob deconv_direct = 1 (one-launch GEMM,
epilogue from the registers), 2 (the same on 128-row tiles, two workgroups per CU), 3 (patch-resident kernel, csrc/deconv_pipe.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

SHAPES = {"deconv1to0": (32, 256, 256, 128, 128), "deconv2to1": (32, 128, 128, 256, 256), "x_192": (32, 128, 128, 192, 192),
          "x_256": (32, 64, 64, 256, 256), "b4_deconv1to0": (4, 256, 256, 128, 128)}
dev = torch.device("cuda", 0)
reps, rounds = int(os.environ.get("GB_REPS", "5")), int(os.environ.get("GB_ROUNDS", "3"))
_lib.load()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for nm, (B, H, W, ci, co) in SHAPES.items():
    x = ops.to_split32(ops.Act(torch.rand(B, H, W, ci, device=dev)))
    w = ops.pack_deconv((np.random.default_rng(0).standard_normal((3, 3, co, ci)) * 0.03).astype(np.float32), dev)
    s, t = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    outs = {k: ops.Act.empty(B, 2 * H, 2 * W, co, dev) for k in (1, 3)}

    def mk(k):
        def f():
            _lib.knob("deconv_direct", k)
            ops.deconv3x3s2_fused(x, w, s, t, outs[k])
        return f

    fns = {k: mk(k) for k in (1, 3)}
    for ew in (4,):
        def mke(ew=ew):
            def f():
                _lib.knob("deconv_direct", 3)
                _lib.knob("epi_width", ew)
                ops.deconv3x3s2_fused(x, w, s, t, outs[3])
                _lib.knob("epi_width", 0)
            return f
        fns[f"3/epi{ew}"] = mke()
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    rel = float((outs[1].buf - outs[3].buf).norm() / outs[1].buf.norm())
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            T[k].append(timed(f))
    fl = 6.0 * B * H * W * 9 * ci * co
    print(f"{nm:14s} rel diff(3 vs 1) {rel:.1e}: " + "  ".join(f"deconv_direct={k} {np.median(T[k]):8.1f} us ({fl / np.median(T[k]) / 1e6 / 25:4.1f}%)" for k in fns), flush=True)
_lib.knob("deconv_direct", 3)
