"""Dev tool: where do sep_pipe2.hip and sep_pipe.hip differ on one shape?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import _lib, ops
dev = torch.device("cuda", 0)
B, H, W, ci, co, res = [int(v) for v in os.environ.get("SD_SHAPE", "1,8,64,64,160,1").split(",")]
g = torch.Generator(device=dev).manual_seed(1)
x = ops.Act(torch.rand(B, H, W, ci, device=dev, generator=g))
w = torch.rand(9, ci, device=dev, generator=g) - 0.5
pw = ops.PackedWeights(np.random.default_rng(0).standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
s1, t1 = torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) - 0.5
r = ops.Act(torch.rand(B, H, W, co, device=dev)) if res else None
outs = {}
for v in (0, 2):
    for rep in range(3):
        _lib.knob("sep_pipe2", v)
        o = ops.Act.empty(B, H, W, co, dev); o.buf.fill_(float("nan"))
        ops.sep_fused(x, w, pw, s1, t1, o, res=r)
        torch.cuda.synchronize()
        outs[(v, rep)] = o.buf.clone()
_lib.knob("sep_pipe2", 0)
ref = outs[(0, 0)]
for k, o in outs.items():
    d = (o != ref) | torch.isnan(o)
    n = int(d.sum())
    print(k, "differs in", n, "values")
    if n:
        idx = d.nonzero()
        print("  image", idx[:, 0].unique().tolist()[:8], "rows", idx[:, 1].unique().tolist()[:40], "\n  cols", idx[:, 2].unique().tolist()[:70], "\n  chans", idx[:, 3].unique().tolist()[:70])
        b, y, xx, c = idx[0].tolist()
        print("  first:", idx[0].tolist(), float(o[b, y, xx, c]), "vs", float(ref[b, y, xx, c]), " max abs diff", float((o - ref).abs().nan_to_num(9e9).max()))
