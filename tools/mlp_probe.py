"""Times the fused Mlp kernels alone on the stage-0 / stage-1 shapes of phi = l, bs 8 (diagnostic).
    VRNET_HIP_LIB=asy-vrnet_amd/csrc/libvrnet_hip_tuning.so VRNET_MLP_DBG=1 python tools/mlp_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asy_vrnet_amd import hip  # noqa: E402

SHAPES = [(131072, 64, 512), (32768, 128, 1024)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M, C, hid in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, C, device="cuda", generator=g)
    res = torch.randn(M, C, device="cuda", generator=g)
    w1 = torch.randn(hid, C, device="cuda", generator=g) / C ** 0.5
    w2 = torch.randn(C, hid, device="cuda", generator=g) / hid ** 0.5
    b1, b2, ls = torch.randn(hid, device="cuda"), torch.randn(C, device="cuda"), torch.rand(C, device="cuda")
    y, u = torch.empty(M, C, device="cuda"), torch.empty(M, hid, device="cuda")
    h, du, dx = torch.empty(M, hid, device="cuda"), torch.empty(M, hid, device="cuda"), torch.empty(M, C, device="cuda")
    pairs = torch.empty(M // 32, C // 32, 2, dtype=torch.float64, device="cuda")
    for prec in (2, 1):
        fwd, bwd = hip.mlp_pack(w1, w2, C, hid, prec)
        tf = timeit(lambda: hip.mlp_fwd(x, C, fwd, b1, b2, res, C, ls, y, C, u, hid, pairs, M, C, hid, prec))
        tn = timeit(lambda: hip.mlp_fwd(x, C, fwd, b1, b2, res, C, ls, y, C, None, 0, None, M, C, hid, prec))
        tb = timeit(lambda: hip.mlp_bwd(res, C, ls, bwd, u, hid, h, hid, du, hid, dx, C, M, C, hid, prec))
        fl = 4.0 * M * C * hid
        print(f"M{M} C{C} H{hid} prec {prec} dbg {os.environ.get('VRNET_MLP_DBG', '0')}: fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s, "
              f"{(3 * M * C + M * hid) * 4 / tf / 1e6:5.2f} TB/s)  fwd(no u, no stats) {tn:7.1f} us ({fl / tn / 1e6:6.1f} TF/s)  "
              f"bwd {tb:7.1f} us ({fl / tb / 1e6:6.1f} TF/s, {(2 * M * C + 3 * M * hid) * 4 / tb / 1e6:5.2f} TB/s)", flush=True)
