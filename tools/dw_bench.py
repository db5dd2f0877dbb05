"""Dev tool: time emd_dw3x3_f32 on the standalone depthwise shapes of graph D."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdenoise import ops
SHAPES = {"512_128": (32, 512, 512, 128, 1), "512_64": (32, 512, 512, 64, 1), "256_384": (32, 256, 256, 384, 1),
          "128_256": (32, 128, 128, 256, 1), "32_728": (32, 32, 32, 728, 1), "512_64_s2": (32, 512, 512, 64, 2)}
dev = torch.device("cuda", 0)
for nm in (sys.argv[1].split(",") if len(sys.argv) > 1 else SHAPES):
    B, H, W, C, s = SHAPES[nm]
    x = ops.Act(torch.rand(B, H, W, C, device=dev)); w = torch.rand(9, C, device=dev)
    split = os.environ.get("DWB_SPLIT") == "1" and s == 1
    out = (ops.SplitAct(B, H, W, C, dev) if split else ops.Act.empty(B, -(-H // s), -(-W // s), C, dev))
    run = (lambda: ops.dw3x3_split32(x, w, out)) if split else (lambda: ops.dw3x3(x, w, out, stride=s))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    by = 4.0 * C * (B * H * W + B * (-(-H // s)) * (-(-W // s)))
    print(f"TH={os.environ.get('EMD_DW_TH','8'):>2s} {nm:10s}: {us:8.1f} us  {by/us/1e3:7.1f} GB/s algorithmic ({by/us/1e3/8000*100:4.1f}% of 8 TB/s)", flush=True)
