"""Dev tool: emd_sep3x3_gemm_f32 on the BASELINE shape ([32,32,32,728] -> 728, with / without residual) against the two-kernel route,
and its per-phase s_memtime sums (emd_debug_sepgemm_stamps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from emdenoise import _lib, ops
dev = torch.device("cuda", 0)
lib = _lib.load()
B = int(os.environ.get("SG_B", "32"))
rng = np.random.default_rng(0)
x = ops.Act(torch.rand(B, 32, 32, 728, device=dev) * 2)
dw = torch.from_numpy((rng.standard_normal((9, 728)) * 0.3).astype(np.float32)).to(dev)
pw = ops.PackedWeights((rng.standard_normal((1, 728, 728)) * 0.05).astype(np.float32), False, dev)
one, zero = torch.ones(728, device=dev), torch.zeros(728, device=dev)
out = ops.Act.empty(B, 32, 32, 728, dev)
res = ops.Act(torch.rand(B, 32, 32, 728, device=dev))
def timeit(f, n=30):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for r in (None, res):
    a = timeit(lambda: ops.sep_gemm(x, dw, pw, one, zero, out, res=r))
    b = timeit(lambda: ops.sep_split32(x, dw, pw, one, zero, out, res=r))
    print(f"B={B} res={r is not None}: sep_gemm {a:.1f} us   dw_split32 + conv1x1_split32 {b:.1f} us")
nwg = B * 8 * 2
st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
lib.emd_debug_sepgemm_stamps(C.c_void_p(st.data_ptr()))
ops.sep_gemm(x, dw, pw, one, zero, out)
torch.cuda.synchronize()
lib.emd_debug_sepgemm_stamps(C.c_void_p(0))
s = st.cpu().numpy().reshape(nwg, 8)[:, :5].astype(np.float64)
tot = s.sum(1)
print("per-workgroup cycles over 23 K steps (median over workgroups): own-DMA wait %.0f  barrier a %.0f  depthwise %.0f  barrier b %.0f  frags+issue+MFMA %.0f  total %.0f (%.0f per K step)"
      % (*np.median(s, 0), np.median(tot), np.median(tot) / 23))
