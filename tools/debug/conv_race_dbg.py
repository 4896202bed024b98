"""Debug: is one conv launch bit-repeatable while another stream keeps the chip busy?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from asy_vrnet_amd import hip
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
SHAPES = [(8, 128, 128, 64, 256), (8, 128, 128, 64, 512), (8, 128, 128, 512, 64), (8, 64, 64, 128, 1024), (8, 64, 64, 1024, 128),
          (8, 32, 32, 320, 1280), (8, 32, 32, 1280, 320), (8, 32, 32, 320, 512), (8, 16, 16, 512, 2048), (8, 16, 16, 2048, 512)]
def mk(B, H, W, ci, co):
    x = torch.randn(B, H, W, ci, device="cuda"); w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
    b = torch.randn(co, device="cuda"); y = torch.empty(B, H, W, co, device="cuda")
    return x, w, b, y
for mode in (0, 1):
  for (B, H, W, ci, co) in SHAPES:
    x, w, b, y = mk(B, H, W, ci, co)
    if mode == 1:
        dy = torch.randn(B, H, W, co, device="cuda"); dx = torch.empty(B, H, W, ci, device="cuda")
    bx, bw, bb_, by = mk(8, 64, 64, 256, 256)
    torch.cuda.synchronize()
    def launch():
        if mode == 0:
            hip.conv2d(x, ci, w, b, y, co, B, H, W, ci, H, W, co, 1, 1, 1, 0, 1, mode=0, precision=prec)
            return y
        hip.conv2d(dy, co, w, None, dx, ci, B, H, W, ci, H, W, co, 1, 1, 1, 0, 1, mode=1, precision=prec)
        return dx
    with torch.cuda.stream(sA):
        ref = launch().clone()
    torch.cuda.synchronize()
    bad = 0
    for it in range(150):
        with torch.cuda.stream(sB):
            for _ in range(3):
                hip.conv2d(bx, 256, bw, bb_, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=prec)
        with torch.cuda.stream(sA):
            out = launch()
            if it % 3 == 0:
                hip.conv2d(bx, 256, bw, bb_, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=0)
            eq = torch.equal(out, ref)
        torch.cuda.synchronize()
        bad += 0 if eq else 1
    print(f"mode {mode} B{B} {H}x{W} {ci}->{co} precision {prec} kernel {hip.last_kernel()}: {bad} of 150 launches differ")
