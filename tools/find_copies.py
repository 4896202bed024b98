"""Diagnostic: which host call sites issue device-to-device copies during one forward+backward step."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asy_vrnet_amd as A
from oracle import vrnet_oracle as O

m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=0)
x, r = A.synthetic_inputs(8, 512, 1, "cuda")
for _ in range(2):
    m.zero_grad(set_to_none=True)
    det, seg = m(x, r)
    O.synthetic_loss(det, seg).backward()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    m.zero_grad(set_to_none=True)
    det, seg = m(x, r)
    O.synthetic_loss(det, seg).backward()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy"):
        st = [s for s in e.stack if "asy" in s or "oracle" in s][:3]
        cnt[(e.name, tuple(st))] += 1
for (k, st), v in cnt.most_common(25):
    print(v, k, " <- ".join(st))
for k in prof.key_averages():
    if any(t in k.key.lower() for t in ("copy", "memcpy", "memset", "fill", "aten::")):
        print(f"{k.count:6d} {k.device_time_total/1e3:9.3f} ms  {k.key[:90]}")
