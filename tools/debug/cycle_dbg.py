import os, sys, gc, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
net = A.EfficientVRNet(4, 9, "nano", img_size=128).cuda().train()
x, r = A.synthetic_inputs(2, 128, 11, "cuda")
with torch.no_grad():
    net(x, r)
gc.collect()
gc.set_debug(gc.DEBUG_SAVEALL)
with torch.no_grad():
    det, seg = net(x, r)
del det, seg
n = gc.collect()
print("collected", n, "objects in cycles")
cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
print(cnt.most_common(12))
for o in gc.garbage:
    if type(o).__name__ == "function":
        print("function", o.__qualname__, o.__code__.co_filename.split("/")[-1], o.__code__.co_firstlineno, "freevars", o.__code__.co_freevars[:12])
