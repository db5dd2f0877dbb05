"""Dev tool: the narrow dense 3x3 convs of graph X's decoder (modified_Xception.py:538-621) -- the tap-major split32 GEMM
(gemm_split_conv_kernel<64>, dev knob conv3_pipe = 0) against the patch-resident kernel (csrc/conv3_pipe.hip).  % = issued bf16 flops / 2.5 PF."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

SHAPES = {"x_dec_128_64": (32, 512, 512, 128, 64), "x_dec_64_64": (32, 512, 512, 64, 64), "b4_64_64": (4, 512, 512, 64, 64),
          "x_dec_192_128": (32, 256, 256, 192, 128), "x_dec_128_128": (32, 256, 256, 128, 128), "x_dec_256_192": (32, 128, 128, 256, 192),
          "x_dec_192_192": (32, 128, 128, 192, 192), "x_dec_384_256": (32, 64, 64, 384, 256), "x_dec_256_256": (32, 64, 64, 256, 256)}
WIDE = int(os.environ.get("C3_WIDE", "2"))
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
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((9, ci, co)) * 0.03).astype(np.float32), False, dev)
    s, t = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    outs = {k: ops.Act.empty(B, H, W, co, dev) for k in (0, 1, 4)}

    def mk(k):
        def f():
            _lib.knob("conv3_pipe", WIDE if k else 0)
            _lib.knob("epi_width", 4 if k == 4 else 1)       # k = 4: the patch-resident kernel with the older 16-byte epilogue
            ops.conv3x3_split32(x, w, s, t, outs[k], act=ops.ACT_RELU, scale2=s, shift2=t)
            _lib.knob("epi_width", 0)
        return f

    fns = {k: mk(k) for k in (0, 1, 4)}
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    rel = float((outs[0].buf - outs[1].buf).norm() / outs[0].buf.norm())
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            T[k].append(timed(f))
    fl = 6.0 * B * H * W * 9 * ci * co
    print(f"{nm:14s} rel diff {rel:.1e}: " + "  ".join(f"conv3_pipe={k} {np.median(T[k]):8.1f} us ({fl / np.median(T[k]) / 1e6 / 25:4.1f}%)" for k in fns), flush=True)
_lib.knob("conv3_pipe", 1)
