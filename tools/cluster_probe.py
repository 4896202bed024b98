"""Times the Cluster kernels alone at the benchmark's shapes (phi = l, bs 8, 512 px): us per launch and TB/s on the
algorithmic bytes (3 P E D 4 forward, 5 P E D 4 backward).   usage: python tools/cluster_probe.py"""
import importlib, sys, torch
sys.path.insert(0, ".")
hip = importlib.import_module("asy-vrnet_amd.hip")


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


tf = tb = 0.0
# (B, H, W, E, D, fold, launches per step)
SHAPES_512 = ((8, 128, 128, 4, 32, 8, 4), (8, 64, 64, 4, 32, 4, 4), (8, 32, 32, 8, 32, 2, 12), (8, 16, 16, 8, 32, 1, 4),
              (8, 64, 64, 4, 24, 2, 1), (8, 32, 32, 4, 24, 2, 1), (8, 16, 16, 4, 24, 2, 1))
# `python tools/cluster_probe.py 1024`: BASELINE configs[4] (1024 px, bs 4): every region has 1 024 points (neck p3: 4 096)
SHAPES_1024 = ((4, 256, 256, 4, 32, 8, 4), (4, 128, 128, 4, 32, 4, 4), (4, 64, 64, 8, 32, 2, 12), (4, 32, 32, 8, 32, 1, 4),
               (4, 128, 128, 4, 24, 2, 1), (4, 64, 64, 4, 24, 2, 1), (4, 32, 32, 4, 24, 2, 1))
for B, H, W, E, D, fold, cnt in (SHAPES_1024 if "1024" in sys.argv[1:] else SHAPES_512):
    C = E * D
    f, v, g = (torch.randn(B, H, W, C, device="cuda") for _ in range(3))
    out, df, dv = (torch.empty(B, H, W, C, device="cuda") for _ in range(3))
    idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    al, be = torch.tensor([1.3], device="cuda"), torch.tensor([-0.2], device="cuda")
    dab = torch.zeros(2, device="cuda")
    fw = timeit(lambda: hip.cluster_fwd(f, v, C, al, be, out, C, idx, wgt, B, H, W, E, D, fold))
    bw = timeit(lambda: hip.cluster_bwd(f, v, C, al, be, idx, g, C, df, dv, C, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold))
    byt = B * H * W * C * 4
    extra = ""
    nst = hip.cluster_state_floats(B, H, W, E, fold)
    if nst:      # regions of more than 256 points: the backward from the forward's saved state (what the step program runs)
        st = torch.empty(nst, device="cuda")
        hip.cluster_fwd(f, v, C, al, be, out, C, idx, wgt, B, H, W, E, D, fold, state=st)
        bw4, bw = bw, timeit(lambda: hip.cluster_bwd(f, v, C, al, be, idx, g, C, df, dv, C, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold,
                                                     saved=(wgt, st)))
        extra = f"   (four-pass backward {bw4:7.1f} us)"
    print(f"{H}x{W} E{E} D{D} fold{fold} x{cnt}: fwd {fw:7.1f} us {3 * byt / fw * 1e-6:6.2f} TB/s   bwd (+ab reduce) {bw:7.1f} us {5 * byt / bw * 1e-6:6.2f} TB/s{extra}", flush=True)
    tf += cnt * fw
    tb += cnt * bw
print(f"per step: forward {tf * 1e-3:.3f} ms, backward {tb * 1e-3:.3f} ms")
