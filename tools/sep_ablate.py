"""Dev tool: phase ablations of sep_pipe.hip (knob sep_ablate: results are wrong on purpose, only the times mean something)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import _lib, ops
SHAPES = {"deconv0_b": (512, 64, 64, 1), "deconv0_a": (512, 128, 64, 0), "cnn1": (256, 128, 128, 0), "deconv1_a": (256, 384, 128, 0)}
dev = torch.device("cuda", 0)
B = 32
_lib.load()
for nm in (os.environ.get("SB_SHAPES", "deconv0_a,deconv1_a,deconv0_b").split(",")):
    S, ci, co, res = SHAPES[nm]
    x = ops.Act(torch.rand(B, S, S, ci, device=dev)); w = torch.rand(9, ci, device=dev) - 0.5
    pw = ops.PackedWeights(np.random.default_rng(0).standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
    s1, t1 = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    out = ops.Act.empty(B, S, S, co, dev)
    r = ops.Act(torch.rand(B, S, S, co, device=dev)) if res else None
    for mode in (0, 1):
        _lib.knob("sep_mode", mode)
        for abl, label in ((0, "full"), (1, "-stage1"), (2, "-mfma"), (4, "-epilogue"), (3, "-stage1-mfma"), (7, "-all compute (DMA only)"), (8, "-patch DMA"),
                           (24, "-patch-weight DMA"), (31, "nothing but barriers"), (32, "-residual loads"), (15, "weight DMA + barriers")):
            _lib.knob("sep_ablate", abl)
            f = lambda: ops.sep_fused(x, w, pw, s1, t1, out, res=r)
            f(); f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            print(f"{nm:10s} mode {mode} {label:26s}: {e0.elapsed_time(e1) * 200:8.1f} us", flush=True)
        _lib.knob("sep_ablate", 0)
_lib.knob("sep_mode", -1)
