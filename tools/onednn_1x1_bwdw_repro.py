"""Dev tool: the oneDNN primitive the float32 oracle tower dies after on the GPU boxes host (EPYC 9575F, 16 threads): backward_weights of
the 1 -> 128 stride-2 1x1 convolution at 512 px (jit_1x1:avx512_core), alone.  CPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ORACLE_MKLDNN"]="1"
import numpy as np, torch
torch.set_num_threads(16)
import torch.nn.functional as F
# the primitive the box dies after: backward_weights of a 1x1 stride-2 conv, ic=1 -> oc=128, 512 -> 256, float32
x = torch.rand(1,1,512,512, requires_grad=False)
w = torch.rand(128,1,1,1, requires_grad=True)
b = torch.rand(128, requires_grad=True)
y = F.conv2d(x, w, b, stride=2)
g = torch.autograd.grad(y.square().sum(), [w, b])
print("ok", g[0].shape, float(g[0].abs().sum()))
for _ in range(20):
    y = F.conv2d(x, w, b, stride=2); g = torch.autograd.grad(y.square().sum(), [w, b])
print("ok x20")
