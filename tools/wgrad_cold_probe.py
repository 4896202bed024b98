"""Weight-gradient launches with COLD operands (rotating through buffer sets larger than the 256 MB MALL), as inside the step:
    python tools/wgrad_cold_probe.py   (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip

SHAPES = [(32768, 256, 256), (8192, 320, 1280), (8192, 1280, 320), (32768, 128, 128), (8192, 320, 256), (8192, 512, 320),
          (131072, 64, 512), (131072, 512, 64), (2048, 512, 512), (2048, 256, 256)]
for M, N, K in SHAPES:
    B, H, W = 8, M // 8 // 64, 64
    per = (M * (N + K)) * 4
    sets = max(2, min(16, int(600e6 // per) + 1))
    xs = [torch.randn(B, H, W, K, device="cuda") for _ in range(sets)]
    dys = [torch.randn(B, H, W, N, device="cuda") for _ in range(sets)]
    dw, db = torch.empty(N, K, 1, 1, device="cuda"), torch.empty(N, device="cuda")
    for warm in (True, False):
        reps = 24
        for i in range(4):
            hip.conv2d_wgrad(xs[0], K, dys[0], N, dw, db, None, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, precision=2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            j = 0 if warm else i % sets
            hip.conv2d_wgrad(xs[j], K, dys[j], N, dw, db, None, B, H, W, K, H, W, N, 1, 1, 1, 0, 1, precision=2)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"wgrad M{M} N{N} K{K} {'warm' if warm else f'cold ({sets} sets)'}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF/s "
              f"operands {per / 1e6:.0f} MB -> {per / us / 1e6:.2f} TB/s", flush=True)
