"""Dev tool: the wide single-output fused separable conv (EMD_SEP_WIDE=1: 128 < Cout <= 256 on 4 x 16 tiles) against the two-kernel route."""
import os, sys
os.environ["EMD_SEP_WIDE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import ops
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for nm, (B, S, ci, co, res) in {"deconv2_a": (32, 128, 384, 256, False), "deconv2_b": (32, 128, 256, 256, True), "cnn2": (32, 128, 128, 256, False),
                                "small": (2, 16, 64, 160, True)}.items():
    x = ops.Act(torch.rand(B, S, S, ci, device=dev)); w = torch.rand(9, ci, device=dev) * 0.3
    pw = ops.PackedWeights((rng.standard_normal((1, ci, co)) * (2.0 / (ci + co)) ** 0.5).astype(np.float32), False, dev)
    s1, t1 = torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) - 0.5
    r = ops.Act(torch.rand(B, S, S, co, device=dev)) if res else None
    assert ops.sep_fused_supported(x, co, 1, 1)
    o1, o2 = ops.Act.empty(B, S, S, co, dev), ops.Act.empty(B, S, S, co, dev)
    fused = lambda: ops.sep_fused(x, w, pw, s1, t1, o1, res=r)
    def pair():
        if ops.conv1x1_split32_supported(B * S * S, ci, co):
            return ops.sep_split32(x, w, pw, s1, t1, o2, res=r)
        tmp = ops.Act.empty(B, S, S, ci, dev)
        ops.dw3x3(x, w, tmp)
        return ops.conv1x1(tmp, pw, s1, t1, o2, res=r)
    fused(); pair(); torch.cuda.synchronize()
    rel = float((o1.buf - o2.buf).norm() / o2.buf.norm())
    print(f"{nm:10s} [{B},{S},{S},{ci}] -> {co}: fused-wide {timed(fused):8.1f} us   two kernels {timed(pair):8.1f} us   rel diff {rel:.1e}", flush=True)
