"""Are the narrow conv kernels repeatable beside other streams? (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asy_vrnet_amd import hip
torch.manual_seed(0)
B, H, W, K, ctot = 8, 64, 64, 256, 9
x = torch.randn(B, H, W, K, device="cuda")
d = torch.randn(B, H, W, ctot, device="cuda")
w = torch.randn(4, K, 1, 1, device="cuda") / 16
bx, bw = torch.randn(8, 64, 64, 256, device="cuda"), torch.randn(256, 256, 1, 1, device="cuda") / 16
by = torch.empty(8, 64, 64, 256, device="cuda")
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
ref = None
bad = [0, 0, 0]
for it in range(300):
    with torch.cuda.stream(s1):
        for _ in range(2):
            hip.conv2d(bx, 256, bw, None, by, 256, 8, 64, 64, 256, 64, 64, 256, 1, 1, 1, 0, 1, mode=0, precision=2)
    with torch.cuda.stream(s3):
        dw3, db3 = torch.empty(1, K, 1, 1, device="cuda"), torch.empty(1, device="cuda")
        hip.conv2d_wgrad(x, K, d[..., 4:], ctot, dw3, db3, None, B, H, W, K, H, W, 1, 1, 1, 1, 0, 1)
    with torch.cuda.stream(s2):
        dw, db = torch.empty(4, K, 1, 1, device="cuda"), torch.empty(4, device="cuda")
        hip.conv2d_wgrad(x, K, d[..., 5:], ctot, dw, db, None, B, H, W, K, H, W, 4, 1, 1, 1, 0, 1)
        dx = torch.empty(B, H, W, K, device="cuda")
        hip.conv2d(d[..., 5:], ctot, w, None, dx, K, B, H, W, K, H, W, 4, 1, 1, 1, 0, 1, mode=1)
        y = torch.empty(B, ctot, H, W, device="cuda")
        hip.conv2d(x, K, w, db, y, 0, B, H, W, K, H, W, 4, 1, 1, 1, 0, 1, out_nchw=1, out_ctot=ctot, out_coff=5)
    torch.cuda.synchronize()
    cur = (dw.clone(), dx.clone(), y[:, 5:].clone())
    if ref is None:
        ref = cur
    else:
        for i in range(3):
            bad[i] += int(not torch.equal(cur[i], ref[i]))
print("differing launches of 299: wgrad", bad[0], "dgrad", bad[1], "fwd", bad[2])
