"""Runs one conv shape a few times (for rocprofv3 --pmc on a single kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip
B, H, W, Ci, Co = [int(v) for v in sys.argv[1:6]]
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
x = torch.randn(B, H, W, Ci if mode == 0 else Co, device="cuda")
w = torch.randn(Co, Ci, 1, 1, device="cuda") * 0.05
y = torch.empty(B, H, W, Co if mode == 0 else Ci, device="cuda")
for _ in range(6):
    hip.conv2d(x, x.shape[-1], w, None, y, y.shape[-1], B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=mode)
torch.cuda.synchronize()
