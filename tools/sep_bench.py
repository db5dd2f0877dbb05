"""Dev tool: A/B of the fused separable conv kernels on graph D's shapes -- sep_fused.hip (register-staged loader) against sep_pipe.hip
(LDS-DMA ring; knobs sep_mode = 0 / 1, sep_nw = 8 / 4) -- with a bit-identity check between them.  SB_STAMPS=1 prints the in-kernel phase split,
SB_SHAPES=a,b picks shapes, SB_B the batch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

# name: (S, Cin, Cout, residual, Cout2 (dual) or 0, split32 output)
SHAPES = {
    "cnn0_last": (512, 64, 64, 0, 0, 0), "deconv0_b": (512, 64, 64, 1, 0, 0), "deconv0_a": (512, 128, 64, 0, 0, 0),
    "deconv0_dual": (512, 128, 64, 0, 64, 0), "cnn1": (256, 128, 128, 0, 0, 0), "deconv1_b": (256, 128, 128, 1, 0, 1),
    "deconv1_a": (256, 384, 128, 0, 0, 0), "deconv1_dual": (256, 384, 128, 0, 128, 0), "cnn2": (128, 128, 256, 0, 0, 0),
    "cnn2_last": (128, 256, 256, 1, 0, 0), }
dev = torch.device("cuda", 0)
B = int(os.environ.get("SB_B", "32"))
lib = _lib.load()
names = os.environ.get("SB_SHAPES", "").split(",") if os.environ.get("SB_SHAPES") else list(SHAPES)
REP = int(os.environ.get("SB_REP", "5"))


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


for nm in names:
    S, ci, co, res, co2, osplit = SHAPES[nm]
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops.Act(torch.rand(B, S, S, ci, device=dev, generator=g))
    w = torch.rand(9, ci, device=dev, generator=g) - 0.5
    rng = np.random.default_rng(0)
    pw = ops.PackedWeights(rng.standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
    s1, t1 = torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) - 0.5
    r = ops.Act(torch.rand(B, S, S, co, device=dev)) if res else None
    if co2:
        pw2 = ops.PackedWeights(rng.standard_normal((1, ci, co2)).astype(np.float32) * 0.1, False, dev)
        sb, tb = torch.rand(co2, device=dev) + 0.5, torch.rand(co2, device=dev) - 0.5
    by = 4.0 * B * S * S * (ci + co * (2 if res else 1) + co2)
    results = {}
    for label, knobs in (("old", {"sep_pipe": 0}), ("lock2", {"sep_pipe": 1, "sep_mode": 0}), ("lock1", {"sep_pipe": 1, "sep_mode": 1}),
                         ("nw4", {"sep_pipe": 1, "sep_mode": 0, "sep_nw": 4}),
                         ("nw4l1", {"sep_pipe": 1, "sep_mode": 1, "sep_nw": 4})):
        _lib.knob("sep_nw", 8)
        if (co2 and label in ("lock2", "nw4", "nw4l1")) or (co > 64 and label.startswith("nw4")) or (
                osplit and label.startswith("nw4")):
            continue
        if os.environ.get("SB_MODES") and label not in os.environ["SB_MODES"].split(","):
            continue
        for k, v in knobs.items():
            _lib.knob(k, v)
        out = ops.SplitAct(B, S, S, co, dev) if osplit else ops.Act.empty(B, S, S, co, dev)
        out.buf.fill_(float("nan"))
        if co2:
            out2 = ops.Act.empty(B, S, S, co2, dev)
            if label == "old" and not ops.sep_dual_supported(x, co, co2):
                continue
            if label == "old" and (co > 64 or co2 > 64):   # the route graph D takes today: two launches
                fn = lambda: (ops.sep_fused(x, w, pw, s1, t1, out), ops.conv1x1(x, pw2, sb, tb, out2))
            else:
                fn = lambda: ops.sep_dual(x, w, pw, pw2, s1, t1, out, sb, tb, out2)
        else:
            fn = lambda: ops.sep_fused(x, w, pw, s1, t1, out, res=r)
        us = timed(fn)
        results[label] = (us, out.buf.clone(), out2.buf.clone() if co2 else None)
        line = f"{nm:13s} {label:6s}: {us:8.1f} us  {by/us/1e3:7.1f} GB/s algorithmic ({by/us/1e3/8000*100:4.1f}% of 8 TB/s)"
        if os.environ.get("SB_STAMPS"):
            st = torch.zeros(B * (S // 8) * (S // 16) * 8, dtype=torch.int64, device=dev)
            lib.emd_debug_sep_stamps(ctypes.c_void_p(st.data_ptr()))
            fn()
            torch.cuda.synchronize()
            lib.emd_debug_sep_stamps(ctypes.c_void_p(0))
            v = st.view(-1, 8).double()
            v = v[v.sum(1) > 0]
            m = v.mean(0)
            tot = m.sum().item()
            nms = (["lds-write+bar1", "issue loads", "depthwise", "bar2", "mfma", "bar3", "epilogue", "-"] if label == "old"
                   else ["waitA+barA", "stage1", "waitB+barB", "issue+mfma", "epilogue", "-", "-", "-"])
            line += f" | stamps/wg {tot:.0f}: " + " ".join(f"{n} {100*a/tot:.0f}%" for n, a in zip(nms, m.tolist()) if a > 0)
        print(line, flush=True)
    base = results.get("old")
    for label in ("lock2", "lock1", "nw4", "nw4l1"):
        if base and label in results:
            same = torch.equal(base[1].view(torch.int32), results[label][1].view(torch.int32))
            same2 = True if base[2] is None else torch.equal(base[2].view(torch.int32), results[label][2].view(torch.int32))
            print(f"{nm:13s} {label} bit-identical to old: {same and same2}", flush=True)
    _lib.knob("sep_pipe", 1)
    _lib.knob("sep_mode", -1)
    _lib.knob("sep_nw", 0)
