"""Host time of one replay of the captured step against its GPU time (diagnostic): is the replay host-bound?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd.graph import GraphedStep

def loss_of(det, seg):
    return sum((d * d).mean() for d in det) + (seg * seg).mean()

dev = torch.device("cuda", 0)
m = A.EfficientVRNet(4, 9, "l", img_size=512).to(dev).train()
A.randomize_state_dict(m.state_dict(), seed=0)
x, r = torch.randn(8, 3, 512, 512, device=dev), torch.rand(8, 4, 512, 512, device=dev)
gs = GraphedStep(m, loss_of, 8, 512, dev)
for _ in range(5):
    gs(x, r)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter()
    gs(x, r)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host per replay: median {sorted(host)[10] * 1e3:.2f} ms, max {max(host) * 1e3:.2f}; all 20 issued in {(t1 - t0) * 1e3:.1f} ms, GPU done after {(t2 - t0) * 1e3:.1f} ms -> {(t2 - t0) * 50:.2f} ms per step")
print("graphs:", len(gs.graphs))

one = []
for _ in range(5):
    torch.cuda.synchronize()
    a = time.perf_counter()
    gs(x, r)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    one.append(((b - a) * 1e3, (c - a) * 1e3))
print("single replay after a sync: (host call ms, until GPU done ms):", [(round(u, 2), round(v, 2)) for u, v in one])
