"""Dev tool: A/B of the stride-2 separable block on graph D's two covered layers (cnn0_strided, cnn1_strided) -- the two-kernel routes
(dw3x3 -> conv1x1, and dw3x3_split32 -> conv1x1_split32 where that applies) against the one-launch form emd_sep3x3_fused_s2_f32
(csrc/sep_pipe.hip, STRIDE = 2; knob sep_mode = 0 / 1) -- with a bit-identity check.  SB_B picks the batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

SHAPES = {"cnn0_strided": (512, 64, 128), "cnn1_strided": (256, 128, 256)}
dev = torch.device("cuda", 0)
B = int(os.environ.get("SB_B", "32"))
REP = int(os.environ.get("SB_REP", "5"))
_lib.load()


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / REP


for nm, (S, ci, co) in SHAPES.items():
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops.Act(torch.rand(B, S, S, ci, device=dev, generator=g))
    w = torch.rand(9, ci, device=dev, generator=g) - 0.5
    pw = ops.PackedWeights(np.random.default_rng(0).standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
    s1, t1 = torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) - 0.5
    So = S // 2
    by = 4.0 * B * (S * S * ci + So * So * co)
    res = {}

    def two():
        tmp = ops.Act.empty(B, So, So, ci, dev)
        ops.dw3x3(x, w, tmp, stride=2)
        return ops.conv1x1(tmp, pw, s1, t1, ops.Act.empty(B, So, So, co, dev))

    def two_split():
        return ops.sep_split32(x, w, pw, s1, t1, ops.Act.empty(B, So, So, co, dev), stride=2)

    routes = [("dw3x3 + conv1x1", two, {})]
    if ops.conv1x1_split32_supported(B * So * So, ci, co):
        routes.append(("split32 pair", two_split, {}))
    for m in (0, 1):
        routes.append((f"one launch, mode {m}", lambda: ops.sep_fused(x, w, pw, s1, t1, ops.Act.empty(B, So, So, co, dev), stride=2),
                       {"sep_mode": m}))
    for label, fn, knobs in routes:
        for k, v in knobs.items():
            _lib.knob(k, v)
        us = timed(fn)
        res[label] = fn().torch().clone()
        print(f"{nm:13s} {label:20s}: {us:8.1f} us  {by/us/1e3:7.1f} GB/s algorithmic ({by/us/1e3/8000*100:4.1f}% of 8 TB/s)", flush=True)
        _lib.knob("sep_mode", -1)
    base = res["dw3x3 + conv1x1"]
    for label, v in res.items():
        if label != "dw3x3 + conv1x1":
            print(f"{nm:13s} {label}: bit-identical to dw3x3 + conv1x1: {torch.equal(base, v)}", flush=True)
