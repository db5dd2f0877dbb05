"""Dev tool: the standalone depthwise launches of graph D at [32,512,512,1], one shape after the other, REPS launches
each -- run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE by tools/dw_traffic.sh, which matches the dispatches to the shapes by order."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdenoise import ops

REPS = 4
# (layer, count in the graph, H = W, channels, stride, rate, split32 output)
SHAPES = [("cnn0_strided", 1, 512, 64, 2, 1, False), ("cnn1_strided", 1, 256, 128, 2, 1, False), ("cnn2", 1, 128, 128, 1, 1, False),
          ("cnn2_last", 1, 128, 256, 1, 1, True), ("cnn2_strided", 1, 128, 256, 2, 1, True), ("cnn3", 1, 64, 256, 1, 1, True),
          ("cnn3_last", 1, 64, 728, 1, 1, True), ("cnn3_strided", 1, 64, 728, 2, 1, True), ("cnn4_* / middle*", 36, 32, 728, 1, 1, True),
          ("aspp_small", 1, 32, 728, 1, 6, True), ("aspp_medium", 1, 32, 728, 1, 12, True), ("aspp_large", 1, 32, 728, 1, 18, True),
          ("deconv2_a", 1, 128, 384, 1, 1, True), ("deconv2_b", 1, 128, 256, 1, 1, True)]

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    B = 32
    rows = []
    for nm, cnt, H, C, s, r, split in SHAPES:
        x = ops.Act(torch.rand(B, H, H, C, device=dev))
        w = torch.rand(9, C, device=dev)
        Ho = -(-H // s)
        out = ops.SplitAct(B, Ho, Ho, C, dev) if split else ops.Act.empty(B, Ho, Ho, C, dev)
        run = (lambda: ops.dw3x3_split32(x, w, out, stride=s, rate=r)) if split else (lambda: ops.dw3x3(x, w, out, stride=s, rate=r))
        run()   # untimed (first use of a kernel loads its code object); tools/dw_traffic.sh drops its counters too
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            run()
        e1.record()
        torch.cuda.synchronize()
        cpad = (C + 31) // 32 * 32 if split else C
        rows.append({"layer": nm, "count": cnt, "H": H, "C": C, "stride": s, "rate": r, "split32_out": split, "us": e0.elapsed_time(e1) * 1e3 / REPS,
                     "algorithmic_bytes": 4.0 * B * (H * H * C + Ho * Ho * cpad)})
    print(json.dumps(rows))
