import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
net = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(net.state_dict(), seed=2)
x, r = A.synthetic_inputs(8, 512, 11, "cuda")
def mem(): return torch.cuda.memory_allocated() / 2**30
for mode in ("no_grad", "record-no-backward", "train"):
    for conc in (1, 0):
        net.concurrent = bool(conc)
        gc.collect(); torch.cuda.synchronize()
        m0 = mem()
        out = []
        for it in range(4):
            if mode == "no_grad":
                with torch.no_grad():
                    det, seg = net(x, r)
            else:
                det, seg = net(x, r)
                if mode == "train":
                    net.zero_grad(set_to_none=True)
                    (seg.mean() + sum(d.mean() for d in det)).backward()
            del det, seg
            torch.cuda.synchronize()
            out.append(round(mem() - m0, 2))
        gc.collect()
        print(f"{mode:20s} concurrent={conc}: GiB held after each of 4 runs {out}; after gc.collect {mem() - m0:.2f}")
