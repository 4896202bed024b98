"""Runs one forward+backward of nano / 512 px with every narrow-kernel conv launch re-run on the MFMA path (tuning library:
VRNET_NARROW toggled in-process) and reports the launches whose results differ."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import asy_vrnet_amd as A
from asy_vrnet_amd import hip
assert hip.tuning_build()
orig = hip.conv2d
bad = 0


def conv2d(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, **kw_):
    global bad
    y0 = y.clone()
    orig(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, **kw_)
    if hip.last_kernel() != 5:
        return
    y1 = y.clone()
    y.copy_(y0)
    os.environ["VRNET_NARROW"] = "0"
    orig(a, lda, w, bias, y, ldy, B, H, W, Cin, OH, OW, Cout, kh, kw, stride, pad, dil, **kw_)
    os.environ["VRNET_NARROW"] = "1"
    err = (y - y1).abs().max().item() / max(y.abs().max().item(), 1e-9)
    tag = f"B{B} {H}x{W} Cin{Cin} Cout{Cout} lda{lda} ldy{ldy} a{tuple(a.shape)}/{a.stride()} y{tuple(y.shape)}/{y.stride()} {kw_.keys()}"
    if err > 1e-4:
        bad += 1
        print("MISMATCH", err, tag, "mode", kw_.get("mode", 0), "acc", kw_.get("accumulate", 0), flush=True)
    else:
        print("ok", f"{err:.1e}", tag, flush=True)
    y.copy_(y1)


hip.conv2d = conv2d
m = A.EfficientVRNet(4, 9, "nano", img_size=512).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=11)
m.concurrent = False
x, r = A.synthetic_inputs(2, 512, 5)
det, seg = m(x.cuda(), r.cuda())
(sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
torch.cuda.synchronize()
print("mismatching launches:", bad)
