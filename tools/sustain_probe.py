"""Diagnostic: does a GEMM launch slow down under sustained load (clock management)?  Same launch repeated; time of
launches 0-19, then after ~0.1 s, ~0.5 s and ~1.5 s of continuous execution.   python tools/sustain_probe.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip

M, N, K = [int(v) for v in sys.argv[1:4]]
precision = int(sys.argv[4]) if len(sys.argv) > 4 else 2
B, W = 16, 64
H = M // B // W
a = torch.randn(B, H, W, K, device="cuda")
w = torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5
y = torch.empty(B, H, W, N, device="cuda")
args = (a, K, w, None, y, N, B, H, W, K, H, W, N, 1, 1, 1, 0, 1)


def burst(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        hip.conv2d(*args, mode=0, precision=precision)
    e1.record()
    return e0, e1, n


hip.conv2d(*args, mode=0, precision=precision)
torch.cuda.synchronize()
import time
time.sleep(0.5)
marks = [burst(20)]
for n in (1000, 20, 4000, 20, 10000, 20):
    marks.append(burst(n))
torch.cuda.synchronize()
t = 0.0
for e0, e1, n in marks:
    ms = e0.elapsed_time(e1)
    print(f"after {t:8.1f} ms of load: {n:6d} launches, {ms * 1e3 / n:7.1f} us each, {2.0 * M * N * K * n / ms / 1e9:6.1f} TF")
    t += ms
