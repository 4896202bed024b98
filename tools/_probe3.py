import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip
m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=0)
m.concurrent = False
x, r = A.synthetic_inputs(8, 512, 1, "cuda")
orig = hip.conv2d_wgrad
log = []
def wrapped(x_, ldx, dy, lddy, dw, db, rs, B, H, W, Cin, OH, OW, Cout, kh, kw, s, p, d, **kw_):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(x_, ldx, dy, lddy, dw, db, rs, B, H, W, Cin, OH, OW, Cout, kh, kw, s, p, d, **kw_)
    e1.record()
    if B * OH * OW == 32768 and Cin == 256 and Cout == 256:
        log.append((e0, e1, ldx, lddy, x_.data_ptr() % 256, dy.data_ptr() % 256, db is None, rs is None, kw_, hip.last_kernel(),
                    float(x_.abs().max()), float(dy.abs().max()), float((dy == 0).float().mean())))
hip.conv2d_wgrad = wrapped
for it in range(3):
    log.clear()
    m.zero_grad(set_to_none=True)
    det, seg = m(x, r)
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
    torch.cuda.synchronize()
for e0, e1, *rest in log:
    print(f"{e0.elapsed_time(e1) * 1e3:7.1f} us", rest)
