"""Dev tool: the pointwise split32 GEMM (gemm_split16_kernel) with its DMA one K step ahead (dev knob split_lead = 1) against two steps
ahead on the same three stages (2), from the B = 32 shapes down to the small-M launches of B = 4 and of the training towers --
interleaved rounds in one process, bit-identity check.  % = issued bf16 flops of the unpadded shape / 2.5 PF."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

SHAPES = {"middle_b32": (32, 32, 32, 728, 728, 0), "middle_res_b32": (32, 32, 32, 728, 728, 1), "half_batch": (16, 32, 32, 728, 728, 0),
          "middle_b8": (8, 32, 32, 728, 728, 0), "middle_b4": (4, 32, 32, 728, 728, 0), "middle_b2": (2, 32, 32, 728, 728, 0),
          "enc64_256_728_b4": (4, 64, 64, 256, 728, 0), "enc64_728_728_b32": (32, 64, 64, 728, 728, 0), "aspp_b4": (4, 32, 32, 2048, 256, 0)}
dev = torch.device("cuda", 0)
reps, rounds = int(os.environ.get("GB_REPS", "20")), int(os.environ.get("GB_ROUNDS", "5"))
_lib.load()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for nm, (B, H, W, ci, co, res) in SHAPES.items():
    x = ops.Act(torch.rand(B, H, W, ci, device=dev) * 2)
    w = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, ci, co)) * 0.05).astype(np.float32), False, dev)
    s, t = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    r = ops.Act(torch.rand(B, H, W, co, device=dev)) if res else None
    xs = ops.to_split32(x)
    outs = {k: ops.Act.empty(B, H, W, co, dev) for k in (1, 2)}

    def mk(k):
        def f():
            _lib.knob("split_lead", k)
            ops.conv1x1_split32(xs, w, s, t, outs[k], res=r)
        return f

    fns = {k: mk(k) for k in (1, 2)}
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    same = bool(torch.equal(outs[1].buf, outs[2].buf))
    T = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            T[k].append(timed(f))
    M = B * H * W
    fl = 6.0 * M * ci * co
    print(f"{nm:18s} M={M:7d} K={ci:4d} N={co:4d} same bits {same}: " + "  ".join(
        f"split_lead={k} {np.median(T[k]):7.1f} us ({fl / np.median(T[k]) / 1e6 / 25:4.1f}%)" for k in fns), flush=True)
_lib.knob("split_lead", 2)
