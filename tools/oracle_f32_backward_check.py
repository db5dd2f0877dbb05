"""One-off check for the round-1 crash record (gpurun_out/train_debug_mc.log: segfault inside the oracle's float32
tower_gradients on the GPU box's host, before any GPU call): run the float32 oracle tower on the CPU with the permuted conv
weight handed to F.conv2d NON-contiguous (the state of oracle/tf_ops.py before commit bdd8cb3) and contiguous (today).
Run under MALLOC_CHECK_=3.  Usage: python tools/oracle_f32_backward_check.py [S] [threads] [noncontig|contig]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
from oracle import tf_ops as T, denoiser_graph as G
from emdenoise import denoiser as D
from tests.synth_inputs import synthetic_pair

S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.set_num_threads(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
mode = sys.argv[3] if len(sys.argv) > 3 else "noncontig"
if mode == "noncontig":
    def conv2d_t(x, w, bias=None, stride=1, rate=1):
        B, H, W, C = x.shape
        kh, kw = w.shape[0], w.shape[1]
        _, pt, pb = T.same_pads(H, kh, stride, rate)
        _, pl, pr = T.same_pads(W, kw, stride, rate)
        xn = F.pad(T._nchw(x), (pl, pr, pt, pb))
        wt = w.permute(3, 2, 0, 1)  # NOT contiguous
        return T._nhwc(F.conv2d(xn, wt, bias, stride=stride, dilation=rate))
    T.conv2d_t = conv2d_t
    if hasattr(G, "conv2d_t"):
        G.conv2d_t = conv2d_t
w = D.synthetic_weights(variant="Dprime")
lq, hq = synthetic_pair(1, S, S, seed=3)
r64 = G.tower_gradients(lq, hq, w, S, dtype=torch.float64)
r32 = G.tower_gradients(lq, hq, w, S, dtype=torch.float32)
a = np.concatenate([r32["grads"][n].ravel().astype(np.float64) for n in r64["grads"]])
b = np.concatenate([r64["grads"][n].ravel() for n in r64["grads"]])
print(f"{mode} S={S} threads={torch.get_num_threads()} MALLOC_CHECK_={os.environ.get('MALLOC_CHECK_')}: f32 loss {r32['loss']:.6f} f64 loss {r64['loss']:.6f} "
      f"grad rel-L2 f32 vs f64 {np.linalg.norm(a-b)/np.linalg.norm(b):.3e}")
