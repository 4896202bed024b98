"""Times single 1x1-conv GEMM launches alone, back to back (host latency does not enter), interleaved rounds in ONE process:
the x6 kernel with pre-split weights that splits the activations itself (igemm_planes_kernel, round 3) against the plane
GEMM on operands that already are bf16 planes (pgemm_kernel, round 4; np = 3 fp32 values, np = 1 bf16 tensors).
usage: python tools/pgemm_probe.py [M N K ...]      (default: the net's shapes at phi = l, bs 8, 512 px)"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
hip = importlib.import_module("asy-vrnet_amd.hip")

DEFAULT = [8192, 320, 1280, 8192, 1280, 320, 8192, 512, 320, 8192, 320, 256, 8192, 2560, 640, 8192, 640, 2560,
           2048, 512, 2048, 2048, 2048, 512, 2048, 512, 512, 2048, 1024, 512, 32768, 256, 256, 32768, 128, 128, 32768, 256, 128,
           32768, 1024, 256, 32768, 256, 1024, 131072, 256, 64, 131072, 128, 128]


def old_planes(w2d, J, K):
    buf = torch.empty((hip.conv_planes_bytes(J, K),), dtype=torch.uint8, device="cuda")
    nb = (K // 16) * 2 * ((J + 127) // 128)
    tab = torch.tensor([w2d.data_ptr(), J, K, K, 1, 0, buf.data_ptr(), 0], dtype=torch.int64, device="cuda")
    hip.conv_planes_pack(tab, 1, nb)
    return buf


def split(x2d, np_):
    R, K = x2d.shape
    out = hip.Planes.empty(np_, (R, K), "cuda")
    hip.planes_from_f32(x2d, K, R, K, out)
    return out


def timeit(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a = [int(v) for v in sys.argv[1:]] or DEFAULT
for i in range(0, len(a), 3):
    M, N, K = a[i:i + 3]
    x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05
    y = torch.empty(M, N, device="cuda")
    pf = old_planes(w, N, K) if hip.conv2d_dma_tile(M, N) else None
    A3, B3, A1, B1 = split(x, 3), split(w, 3), split(x, 1), split(w, 1)
    yp3 = hip.Planes.empty(3, (M, N), "cuda")
    fl = 2.0 * M * N * K
    arms = [("r3 x6, weights pre-split", lambda: hip.conv2d(x, K, w, None, y, N, 1, 1, M, K, 1, M, N, 1, 1, 1, 0, 1, precision=2, w_planes=pf)),
            ("plane GEMM np=3", lambda: hip.gemm_planes(A3, B3, M, N, K, y=y, ldy=N)),
            ("plane GEMM np=3 -> planes", lambda: hip.gemm_planes(A3, B3, M, N, K, yp=yp3)),
            ("plane GEMM np=1 (bf16)", lambda: hip.gemm_planes(A1, B1, M, N, K, y=y, ldy=N)),
            ("fp32 -> planes pass (A)", lambda: hip.planes_from_f32(x, K, M, K, A3))]
    if not hip.gemm_planes_ok(M, N, K):
        arms = arms[:1]
    for _, fn in arms:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    best = {}
    for rnd in range(3):
        for name, fn in arms:
            us = timeit(fn, 30)
            best.setdefault(name, []).append(us)
    for name, _ in arms:
        v = sorted(best[name])
        print(f"M{M} N{N} K{K} {name:28s} median {v[1]:8.1f} us  min {v[0]:8.1f} us  {fl / v[1] * 1e-6:7.1f} TFLOP/s", flush=True)
    # weight gradient of the same layer: contraction over the M rows, x = the A operand (K channels), dy = M x N
    if hip.wgrad_planes_ok(M, K, N):
        dyf = torch.randn(M, N, device="cuda")
        DY3, DY1 = split(dyf, 3), split(dyf, 1)
        dw = torch.empty(N, K, device="cuda")
        warms = [("r3 x6 weight gradient", lambda: hip.conv2d_wgrad(x, K, dyf, N, dw, None, None, 1, 1, M, K, 1, M, N, 1, 1, 1, 0, 1, precision=2)),
                 ("plane weight gradient np=3", lambda: hip.wgrad_planes(A3, DY3, M, K, N, dw)),
                 ("plane weight gradient np=1", lambda: hip.wgrad_planes(A1, DY1, M, K, N, dw))]
        for _, fn in warms:
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        best = {}
        for rnd in range(3):
            for name, fn in warms:
                best.setdefault(name, []).append(timeit(fn, 30))
        for name, _ in warms:
            v = sorted(best[name])
            print(f"M{M} N{N} K{K} {name:28s} median {v[1]:8.1f} us  min {v[0]:8.1f} us  {fl / v[1] * 1e-6:7.1f} TFLOP/s (incl. slab reduce)", flush=True)
