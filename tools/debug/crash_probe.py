import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import torch
import asy_vrnet_amd as A
m = A.EfficientVRNet(4, 9, "nano", img_size=64).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=1)
x, r = A.synthetic_inputs(2, 64, 1)
x, r = x.cuda(), r.cuda()
print("fwd"); sys.stdout.flush()
det, seg = m(x, r)
torch.cuda.synchronize(); print("fwd ok"); sys.stdout.flush()
(sum((d*d).mean() for d in det) + (seg*seg).mean()).backward()
torch.cuda.synchronize(); print("bwd ok")
