"""Dev tool: the weight-gradient GEMM of a pair-sized 728 -> 728 layer (M = 2048), 50 launches, for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emdenoise import _lib, ops, train_ops as TO
dev = torch.device("cuda", 0)
_lib.load()
B, H, K, N = int(os.environ.get("WG_B", "2")), 32, 728, 728
x = ops.Act(torch.randn(B, H, H, K, device=dev))
dy = ops.Act(torch.randn(B, H, H, N, device=dev))
dw = torch.zeros(1, K, N, device=dev)
for _ in range(50):
    TO.conv_wgrad(x, dy, dw, [0], [0])
torch.cuda.synchronize()
print("done")
