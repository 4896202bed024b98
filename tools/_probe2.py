import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip
M, N, K = 32768, 256, 256
B, H, W = 8, 64, 64
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
dw, db = torch.empty(N, K, 1, 1, device="cuda"), torch.empty(N, device="cuda")
for name, scale_dy, zero_frac, scale_x in (("normal", 1.0, 0.0, 1.0), ("dy 1e-9", 1e-9, 0.0, 1.0), ("dy 1e-9 half zeros", 1e-9, 0.5, 1.0),
                                           ("dy 1e-20", 1e-20, 0.0, 1.0), ("dy 1e-30", 1e-30, 0.0, 1.0), ("dy 1e-36", 1e-36, 0.0, 1.0),
                                           ("x relu", 1.0, 0.0, -1.0)):
    x = torch.randn(B, H, W, K, device="cuda")
    if scale_x < 0: x = torch.relu(x)
    dy = torch.randn(B, H, W, N, device="cuda") * scale_dy
    if zero_frac: dy = dy * (torch.rand_like(dy) > zero_frac)
    for db_ in (db, None):
        us = t(lambda: hip.conv2d_wgrad(x, K, dy, N, dw, db_, None, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, precision=2))
        print(f"{name:22s} bias {db_ is not None}: {us:7.1f} us", flush=True)
