"""Diagnostic: Cluster core forward / backward launch time and HBM rate at the four stage shapes of phi-l (bs 8).
    python tools/cluster_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asy_vrnet_amd import hip

SHAPES = [(8, 128, 128, 4, 32, 8), (8, 64, 64, 4, 32, 4), (8, 32, 32, 8, 32, 2), (8, 16, 16, 8, 32, 1)]   # B H W E D fold


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B, H, W, E, D, fold in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    ld = 2 * E * D
    fv = torch.randn(B, H, W, ld, device="cuda", generator=g)
    f, v = fv[..., :E * D], fv[..., E * D:]
    out = torch.empty(B, H, W, E * D, device="cuda")
    idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    al, be = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    dout = torch.randn(B, H, W, E * D, device="cuda", generator=g)
    dfv = torch.empty_like(fv)
    da, db = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    tf = timeit(lambda: hip.cluster_fwd(f, v, ld, al, be, out, E * D, idx, wgt, B, H, W, E, D, fold))
    tb = timeit(lambda: hip.cluster_bwd(f, v, ld, al, be, idx, dout, E * D, dfv[..., :E * D], dfv[..., E * D:], ld, da, db, 0,
                                        B, H, W, E, D, fold))
    pts = B * H * W * E * D * 4
    print(f"B{B} {H}x{W} E{E} D{D} fold{fold}: fwd {tf:7.1f} us {3 * pts / tf / 1e6:6.2f} TB/s | bwd {tb:7.1f} us {5 * pts / tb / 1e6:6.2f} TB/s")
