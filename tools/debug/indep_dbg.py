"""Diagnostic: eval-mode batch independence.  Records every tensor the C-ABI wrappers write during two forwards
that differ only in images 4..7 and prints the first launches whose rows for images 0..3 differ."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip

phi = sys.argv[1] if len(sys.argv) > 1 else "l"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
B = 8
net = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().eval()
A.randomize_state_dict(net.state_dict(), seed=2)
x, r = A.synthetic_inputs(B, size, 12, "cuda")
x2, r2 = A.synthetic_inputs(B, size, 13, "cuda")
x2[:B // 2], r2[:B // 2] = x[:B // 2], r[:B // 2]

log = []
names = [n for n in dir(hip) if callable(getattr(hip, n)) and not n.startswith("_")]


def wrap(name, fn):
    def inner(*a, **k):
        out = fn(*a, **k)
        rec = []
        for i, t in enumerate(list(a) + list(k.values())):
            if torch.is_tensor(t) and t.is_cuda and t.dim() >= 1 and t.shape[0] == B and t.dtype in (torch.float32, torch.uint8, torch.float64):
                rec.append((i, t[:B // 2].clone()))
        log.append((name, rec))
        return out
    return inner


for n in names:
    f = getattr(hip, n)
    if getattr(f, "__module__", "") == hip.__name__ and n not in ("lib", "load"):
        setattr(hip, n, wrap(n, f))
import asy_vrnet_amd.program as P
with torch.no_grad():
    net(x, r)
    log.clear()
    net(x, r)
    la = list(log)
    log.clear()
    net(x2, r2)
    lb = list(log)
print(len(la), len(lb))
shown = 0
for j, ((na, ra), (nb, rb)) in enumerate(zip(la, lb)):
    assert na == nb
    for (i, ta), (_, tb) in zip(ra, rb):
        if not torch.equal(ta, tb):
            d = (ta.double() - tb.double()).abs().max().item()
            print(f"launch {j} {na} arg {i} shape {tuple(ta.shape)} maxdiff {d:.3e} scale {ta.double().abs().max().item():.3e}")
            shown += 1
    if shown > 12:
        break
