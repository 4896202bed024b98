import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd.parallel import DataParallelVRNet
from asy_vrnet_amd.graph import GraphedStep


def loss_of(det, seg):
    return sum((d * d).mean() for d in det) + (seg * seg).mean()


def build(seed):
    m = A.EfficientVRNet(4, 9, "nano", img_size=64).cuda()
    A.randomize_state_dict(m.state_dict(), seed=seed)
    return m.train(True)


def diff(ga, gb):
    bad = [(k, float((ga[k] - gb[k]).abs().max() / gb[k].abs().max().clamp_min(1e-12))) for k in ga if not torch.equal(ga[k], gb[k])]
    return len(bad), bad[:3]


x, r = A.synthetic_inputs(2, 64, 3)
x, r = x.cuda(), r.cuda()
eager = []
for i in range(3):
    m = build(7)
    if len(sys.argv) > 1:
        m.pair_streams = False
    loss_of(*m(x, r)).backward()
    torch.cuda.synchronize()
    eager.append({k: p.grad.clone() for k, p in m.named_parameters() if p.numel()})
    print("eager run", i, "vs run 0:", diff(eager[i], eager[0]))
for segs in ():
    m2 = build(7)
    dp2 = DataParallelVRNet(m2, bucket_bytes=1 << 20, segments=segs)
    gs = GraphedStep(dp2, loss_of, 2, 64, x.device, warmup=2)
    for rep in range(2):
        m2.load_state_dict(build(7).state_dict())
        gs(x, r)
        torch.cuda.synchronize()
        g = {k: p.grad.clone() for k, p in m2.named_parameters() if p.numel()}
        print("segments", segs, "rep", rep, "vs eager0", diff(g, eager[0]), "vs eager2", diff(g, eager[2]))
for conc in (True, False):
    m3 = build(7)
    m3.concurrent = conc
    if os.environ.get("NOPAIR"):
        m3.pair_streams = False
    gs = GraphedStep(m3, loss_of, 2, 64, x.device, warmup=2)
    for rep in range(3):
        m3.load_state_dict(build(7).state_dict())
        gs(x, r)
        torch.cuda.synchronize()
        g = {k: p.grad.clone() for k, p in m3.named_parameters() if p.numel()}
        print("non-DP graph concurrent", conc, "rep", rep, "vs eager0", diff(g, eager[0]))
