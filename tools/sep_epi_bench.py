"""Dev tool: the fused separable conv (sep_pipe.hip) on graph D's shapes, A/B of one dev knob (SEB_KNOB=epi_width: per-channel dword
epilogue 1 against the transposed 16-byte one 4; SEB_KNOB=sep_nw: 8-wave workgroups on 8 x 32 tiles, one per CU, against 4-wave ones on
8 x 16 tiles, two per CU), with a bit-identity check between the two."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

# name: (S, Cin, Cout, residual, Cout2 (dual) or 0, split32 output, stride)
SHAPES = {
    "cnn0_last": (512, 64, 64, 0, 0, 0, 1), "deconv0_b": (512, 64, 64, 1, 0, 0, 1), "deconv0_a": (512, 128, 64, 0, 0, 0, 1),
    "deconv0_dual": (512, 128, 64, 0, 64, 0, 1), "cnn1": (256, 128, 128, 0, 0, 0, 1), "deconv1_b": (256, 128, 128, 1, 0, 1, 1),
    "deconv1_a": (256, 384, 128, 0, 0, 0, 1), "deconv1_dual": (256, 384, 128, 0, 128, 0, 1), "cnn2": (128, 128, 256, 0, 0, 0, 1),
    "cnn2_last": (128, 256, 256, 1, 0, 0, 1), "cnn0_strided": (512, 64, 128, 0, 0, 0, 2), "cnn1_strided": (256, 128, 256, 0, 0, 0, 2)}
dev = torch.device("cuda", 0)
B = int(os.environ.get("SB_B", "32"))
REP, ROUNDS = 5, 3
KNOB = os.environ.get("SEB_KNOB", "epi_width")
VALS = tuple(int(v) for v in os.environ["SEB_VALS"].split(",")) if "SEB_VALS" in os.environ else ((1, 4) if KNOB == "epi_width" else (0, 2) if KNOB == "sep_pipe2" else (8, 4))
RESET = 0
ONLY = os.environ.get("SEB_ONLY", "")
_lib.load()


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / REP


tot = {v: 0.0 for v in VALS}
for nm, (S, ci, co, res, co2, osplit, stride) in SHAPES.items():
    if ONLY and nm not in ONLY.split(","):
        continue
    g = torch.Generator(device=dev).manual_seed(1)
    x = ops.Act(torch.rand(B, S, S, ci, device=dev, generator=g))
    w = torch.rand(9, ci, device=dev, generator=g) - 0.5
    rng = np.random.default_rng(0)
    pw = ops.PackedWeights(rng.standard_normal((1, ci, co)).astype(np.float32) * 0.1, False, dev)
    s1, t1 = torch.rand(co, device=dev) + 0.5, torch.rand(co, device=dev) - 0.5
    So = S // stride
    r = ops.Act(torch.rand(B, So, So, co, device=dev)) if res else None
    if co2:
        pw2 = ops.PackedWeights(rng.standard_normal((1, ci, co2)).astype(np.float32) * 0.1, False, dev)
        sb, tb = torch.rand(co2, device=dev) + 0.5, torch.rand(co2, device=dev) - 0.5
    outs, fns = {}, {}
    for ew in VALS:
        out = ops.SplitAct(B, So, So, co, dev) if osplit else ops.Act.empty(B, So, So, co, dev)
        out.buf.fill_(float("nan"))
        out2 = ops.Act.empty(B, So, So, co2, dev) if co2 else None
        outs[ew] = (out, out2)

        def fn(ew=ew, out=out, out2=out2):
            _lib.knob(KNOB, ew)
            if co2:
                ops.sep_dual(x, w, pw, pw2, s1, t1, out, sb, tb, out2)
            else:
                ops.sep_fused(x, w, pw, s1, t1, out, res=r, stride=stride)
            _lib.knob(KNOB, RESET)
        fns[ew] = fn
    for f in fns.values():
        f(); f()
    torch.cuda.synchronize()
    a, b = VALS
    same = torch.equal(outs[a][0].buf, outs[b][0].buf) and (not co2 or torch.equal(outs[a][1].buf, outs[b][1].buf))
    T = {k: [] for k in fns}
    for _ in range(ROUNDS):
        for k, f in fns.items():
            T[k].append(timed(f))
    for k in T:
        tot[k] += float(np.median(T[k]))
    gb = 4.0 * B * (S * S * ci + So * So * (co + co2 + (co if res else 0))) / 1e9      # algorithmic bytes: input + outputs (+ residual)
    stamp_txt = ""
    if os.environ.get("SEB_STAMPS"):     # in-kernel phase stamps of workgroup wave 0 (sep_pipe2: top of slot = DMA issue + epilogue | slot body | wait + barrier)
        import ctypes
        lib = _lib.load()
        for k, f in [(k, f) for k, f in fns.items()] + ([(("wave4", b), fns[b])] if KNOB == "sep_pipe2" else []):
            if isinstance(k, tuple):
                _lib.knob("sep_stamp_wave", 4)
            st = torch.zeros(B * (S // 8) * (S // 16) * 8, dtype=torch.int64, device=dev)
            lib.emd_debug_sep_stamps(ctypes.c_void_p(st.data_ptr()))
            f()
            torch.cuda.synchronize()
            lib.emd_debug_sep_stamps(ctypes.c_void_p(0))
            _lib.knob("sep_stamp_wave", 0)
            v = st.view(-1, 8).double()
            v = v[v.sum(1) > 0]
            m = v.mean(0)
            stamp_txt += f"\n      stamps {KNOB}={k}: {len(v)} workgroups, {m.sum().item():.0f} ticks each: " + " ".join(f"[{i}] {100 * a / m.sum().item():.0f}%" for i, a in enumerate(m.tolist()) if a > 0)
    print(f"{nm:13s} same bits {same}: " + "  ".join(f"{KNOB}={k} {np.median(T[k]):8.1f} us ({gb / np.median(T[k]) * 1e3:5.2f} TB/s)" for k in T) + stamp_txt, flush=True)
print("sum: " + "  ".join(f"{KNOB}={k} {v:8.1f} us" for k, v in tot.items()))
