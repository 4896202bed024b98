"""Diagnostic: 1x1-conv GEMM shapes of the net at precision 0 (fp32 MFMA) vs 2 (six bf16 products per fp32 product):
time per launch and error against an fp64 matmul.   python tools/x6_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip

SHAPES = [  # M, N, K
    (16384, 320, 1280), (16384, 1280, 320), (262144, 64, 512), (262144, 512, 64), (65536, 128, 1024), (65536, 1024, 128),
    (16384, 512, 320), (8192, 640, 2560), (8192, 2560, 640), (16384, 320, 256), (262144, 256, 64), (32768, 256, 256),
    (4096, 512, 2048), (4096, 2048, 512), (65536, 256, 128), (2048, 512, 512)]


def run(M, N, K, mode, precision, reps=20):
    B, H, W = 16, M // 16 // 64 if M >= 16 * 64 else 1, 64
    if B * H * W != M:
        B, H, W = 1, M // 64, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    ci, co = (K, N) if mode == 0 else (N, K)
    a = torch.randn(B, H, W, K, device="cuda", generator=g)
    w = torch.randn(co, ci, 1, 1, device="cuda", generator=g) / K ** 0.5
    y = torch.empty(B, H, W, N, device="cuda")
    args = (a, K, w, None, y, N, B, H, W, ci, H, W, co, 1, 1, 1, 0, 1)
    for _ in range(3):
        hip.conv2d(*args, mode=mode, precision=precision)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.conv2d(*args, mode=mode, precision=precision)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    rows = slice(0, min(M, 4096))
    a2 = a.view(M, K)[rows].double()
    wm = w.view(co, ci).double()
    ref = a2 @ (wm.t() if mode == 0 else wm)
    err = ((y.view(M, N)[rows].double() - ref).abs().max() / ref.abs().max()).item()
    return us, 2.0 * M * N * K / us / 1e6, err


def run_wgrad(M, N, K, precision, reps=20):
    B, H, W = 16, M // 16 // 64 if M >= 16 * 64 else 1, 64
    if B * H * W != M:
        B, H, W = 1, M // 64, 64
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, H, W, K, device="cuda", generator=g)
    dy = torch.randn(B, H, W, N, device="cuda", generator=g)
    dw, db = torch.empty(N, K, 1, 1, device="cuda"), torch.empty(N, device="cuda")
    args = (x, K, dy, N, dw, db, None, B, H, W, K, H, W, N, 1, 1, 1, 0, 1)
    for _ in range(3):
        hip.conv2d_wgrad(*args, precision=precision)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hip.conv2d_wgrad(*args, precision=precision)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = dy.view(M, N).double().t() @ x.view(M, K).double()
    err = ((dw.view(N, K).double() - ref).abs().max() / ref.abs().max()).item()
    berr = ((db.double() - dy.view(M, N).double().sum(0)).abs().max() / dy.view(M, N).double().sum(0).abs().max()).item()
    return us, 2.0 * M * N * K / us / 1e6, max(err, berr)


if "wgrad" in sys.argv:
    for M, N, K in SHAPES:
        u0, t0, e0 = run_wgrad(M, N, K, 0)
        u2, t2, e2 = run_wgrad(M, N, K, 2)
        print(f"wgrad  M{M:7d} N{N:5d} K{K:5d}: fp32 {u0:8.1f} us {t0:6.1f} TF err {e0:.1e} | x6 {u2:8.1f} us {t2:6.1f} TF err {e2:.1e} | x{u0 / u2:.2f}")
    sys.exit(0)
for mode in (0, 1):
    for M, N, K in SHAPES:
        u0, t0, e0 = run(M, N, K, mode, 0)
        u2, t2, e2 = run(M, N, K, mode, 2)
        tail = ""
        if "bf16" in sys.argv and mode == 0:      # same tiles, one bf16 product per fp32 product: the non-arithmetic floor of the x6 kernel
            u3, t3, e3 = run(M, N, K, mode, 3)
            tail = f" | bf16-on-tile {u3:8.1f} us (bytes {4e-6 * (M * K + M * N + N * K) / u3:5.2f} TB/s)"
        print(f"mode {mode} M{M:7d} N{N:5d} K{K:5d}: fp32 {u0:8.1f} us {t0:6.1f} TF err {e0:.1e} | x6 {u2:8.1f} us {t2:6.1f} TF err {e2:.1e} | x{u0 / u2:.2f}{tail}")
