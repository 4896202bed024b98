"""Times single conv launches (forward / data gradient, with and without pre-split weights) at given shapes, back to back in a
loop so that host latency does not enter.   usage: python tools/planes_shape_probe.py [--only "fwd  x6 planes"] B H W Cin Cout [B H W Cin Cout ...]
PROBE_COLD=1: the launches rotate through operand / output sets that together exceed the 256 MB Infinity Cache (as inside the
step, where a layer's input was written a few launches ago and its weights were last read a step ago)."""
import importlib, os, sys, torch
sys.path.insert(0, ".")
hip = importlib.import_module("asy-vrnet_amd.hip")


def planes(w2d, J, K, sj, sk, kscale=None):
    buf = torch.empty((hip.conv_planes_bytes(J, K),), dtype=torch.uint8, device="cuda")
    nb = (K // 16) * 2 * ((J + 127) // 128)
    tab = torch.tensor([w2d.data_ptr(), J, K, sj, sk, 0 if kscale is None else kscale.data_ptr(), buf.data_ptr(), 0], dtype=torch.int64, device="cuda")
    hip.conv_planes_pack(tab, 1, nb)
    return buf


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


only = None
argv = sys.argv[1:]
if argv and argv[0] == "--only":      # --only "<substring of the row name>"
    only, argv = argv[1], argv[2:]
a = [int(v) for v in argv]
for i in range(0, len(a), 5):
    B, H, W, Ci, Co = a[i:i + 5]
    M = B * H * W
    pad = int(os.environ.get("PROBE_LD_PAD", "0"))      # extra floats per row of the input operands (row-stride experiments)
    cold = os.environ.get("PROBE_COLD", "0") == "1"
    sets = max(2, int(700e6 // (M * (Ci + Co) * 4)) + 1) if cold else 1
    xs = [torch.randn(M, Ci + pad, device="cuda") for _ in range(sets)]
    gs = [torch.randn(M, Co + pad, device="cuda") for _ in range(sets)]
    ys = [torch.empty(M, Co, device="cuda") for _ in range(sets)]
    dxs = [torch.empty(M, Ci, device="cuda") for _ in range(sets)]
    w, ls = torch.randn(Co, Ci, device="cuda") * 0.05, torch.randn(Co, device="cuda")
    aux = torch.randn(M, Ci, device="cuda")
    rot = [0]

    def nxt(lst):
        rot[0] += 1
        return lst[(rot[0] // 2) % sets]

    pf, pb = planes(w, Co, Ci, Ci, 1), planes(w, Ci, Co, 1, Ci)
    fl = 2.0 * M * Ci * Co
    for name, fn in (
        ("fwd  fp32", lambda: hip.conv2d(nxt(xs), Ci + pad, w, None, nxt(ys), Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=0)),
        ("fwd  x6", lambda: hip.conv2d(nxt(xs), Ci + pad, w, None, nxt(ys), Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2)),
        ("fwd  x6 planes", lambda: hip.conv2d(nxt(xs), Ci + pad, w, None, nxt(ys), Co, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, precision=2, w_planes=pf)),
        ("dgrad fp32", lambda: hip.conv2d(nxt(gs), Co + pad, w, None, nxt(dxs), Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, precision=0)),
        ("dgrad x6", lambda: hip.conv2d(nxt(gs), Co + pad, w, None, nxt(dxs), Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, precision=2)),
        ("dgrad x6 planes", lambda: hip.conv2d(nxt(gs), Co + pad, w, None, nxt(dxs), Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, precision=2, w_planes=pb)),
        ("dgrad x6 planes gelu'", lambda: hip.conv2d(nxt(gs), Co + pad, w, None, nxt(dxs), Ci, B, H, W, Ci, H, W, Co, 1, 1, 1, 0, 1, mode=1, aux=aux, ldaux=Ci, precision=2, w_planes=pb)),
    ):
        if only and only not in name:
            continue
        us = timeit(fn)
        print(f"M{M} Cin{Ci} Cout{Co} {name:22s} k{hip.last_kernel()} {us:8.1f} us {fl / us * 1e-6:7.1f} TFLOP/s" + (f"  cold ({sets} sets)" if cold else ""), flush=True)
