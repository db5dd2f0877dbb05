"""Dev tool: the generated-input fused separable conv (graph D's cnn0 -> cnn0_last, [32,512,512,1 -> 64 -> 64]) on sep_pipe.hip's 4-wave
instance (dev knob sep_gen_pipe = 1) against sep_fused.hip's register-staged kernel (0): time and bit identity."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from emdenoise import _lib, ops

dev = torch.device("cuda", 0)
B, S, C = int(os.environ.get("GB_B", "32")), int(os.environ.get("GB_S", "512")), 64
_lib.load()
g = torch.Generator(device=dev).manual_seed(1)
d4 = ops.Act(torch.rand(B, S, S, 4, device=dev, generator=g))
a, t = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
dw = torch.rand(9, C, device=dev) - 0.5
pw = ops.PackedWeights(np.random.default_rng(0).standard_normal((1, C, C)).astype(np.float32) * 0.1, False, dev)
s1, t1 = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
KS = (0, 1, 2)      # 2 = the pipelined kernel with the transposed 16-byte epilogue (dev knob epi_width = 4)
outs, T = {}, {k: [] for k in KS}
for k in KS:
    outs[k] = ops.Act.empty(B, S, S, C, dev)
    outs[k].buf.fill_(float("nan"))


def run(k):
    _lib.knob("sep_gen_pipe", min(k, 1))
    _lib.knob("epi_width", 4 if k == 2 else 0)
    ops.sep_fused_gen(d4, a, t, dw, pw, s1, t1, outs[k], gen_act=1)
    _lib.knob("sep_gen_pipe", 0)
    _lib.knob("epi_width", 0)


for k in KS:
    run(k); run(k)
torch.cuda.synchronize()
print("same bits:", [torch.equal(outs[0].buf, outs[k].buf) for k in KS], "nan:", bool(torch.isnan(outs[1].buf).any()))
for _ in range(5):
    for k in KS:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(k)
        e1.record()
        torch.cuda.synchronize()
        T[k].append(e0.elapsed_time(e1) * 200.0)
gb = 4.0 * B * S * S * (C + 1) / 1e9
for k in KS:
    us = float(np.median(T[k]))
    print(f"sep_gen_pipe={k}: {us:8.1f} us  {gb / us * 1e3:5.2f} TB/s (output + d)")
if os.environ.get("GB_STAMPS"):   # in-kernel phase stamps of the pipelined kernel (wave 0 of every workgroup): [0] wait + barrier A, [1] generate + stage 1,
    import ctypes                 # [2] wait + barrier B, [3] stage 2, [4] epilogue
    lib = _lib.load()
    for k in (1, 2):
        st = torch.zeros(B * (S // 8) * (S // 16) * 8, dtype=torch.int64, device=dev)
        lib.emd_debug_sep_stamps(ctypes.c_void_p(st.data_ptr()))
        run(k)
        torch.cuda.synchronize()
        lib.emd_debug_sep_stamps(ctypes.c_void_p(0))
        v = st.view(-1, 8).double()
        v = v[v.sum(1) > 0]
        m = v.mean(0)
        print(f"stamps k={k}: {len(v)} workgroups, {m.sum().item():.0f} ticks each: " + " ".join(f"[{i}] {100 * x / m.sum().item():.0f}%" for i, x in enumerate(m.tolist()) if x > 0))
