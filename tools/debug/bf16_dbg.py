import os, sys
sys.path.insert(0, os.getcwd())
import torch
import asy_vrnet_amd as A
from oracle import vrnet_oracle as O
from tests.parity import hip_idx_maps, rel_err
phi, size, batch = sys.argv[1], int(sys.argv[2]), 2
m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=21)
x, r = A.synthetic_inputs(batch, size, 31)
sd0 = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
m.compute_dtype = "bf16"
with torch.no_grad():
    det, seg = m(x.cuda(), r.cuda())
forced = hip_idx_maps(m)
P = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd0.items()}
orig = O._round_ste
for name, fx, fw in (("x and w", True, True), ("none", False, False), ("x only", True, False), ("w only", False, True)):
    calls = {"n": 0}
    def conv(P_, pre, x_, stride=1, pad=0, dil=1, groups=1, fx=fx, fw=fw):
        w = P_[pre + ".weight"]
        if groups == 1:
            co, ci, kh, kw = w.shape
            patch = kh == stride and kh > 1 and pad == 0
            ck = ci * kh * kw if patch else ci
            if ck % 4 == 0 and co > 32 and co % 4 == 0:
                if fx: x_ = orig(x_)
                if fw: w = orig(w)
        return torch.nn.functional.conv2d(x_, w, P_.get(pre + ".bias"), stride, pad, dil, groups)
    O.conv = conv
    with torch.no_grad():
        det_o, seg_o, ctx = O.forward(P, x.double(), r.double(), phi, True, forced_idx=forced)
    print(f"{name:8s} det_err {max(rel_err(a, b) for a, b in zip(det, det_o)):.3e} seg_err {rel_err(seg, seg_o):.3e} flips {sum(v.get('mismatch', 0) for v in ctx.idx_report.values())}")
