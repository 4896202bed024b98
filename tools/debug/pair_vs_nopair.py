import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from oracle import vrnet_oracle as O
phi, size, batch, pseed, iseed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
x, r = A.synthetic_inputs(batch, size, iseed)
res = {}
for tag, pair in (("pair", True), ("masked", False)):
    m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
    A.randomize_state_dict(m.state_dict(), seed=pseed)
    m.pair_streams = pair
    if pair:
        os.environ.pop("VRNET_PAIR_MASK", None)
    else:
        os.environ["VRNET_PAIR_MASK"] = sys.argv[6]
    m.concurrent = False
    xg, rg = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
    det, seg = m(xg, rg)
    O.synthetic_loss(det, seg).backward()
    torch.cuda.synchronize()
    res[tag] = ({k: p.grad.clone() for k, p in m.named_parameters() if p.numel()}, xg.grad.clone(), rg.grad.clone(),
                {k: v.clone() for k, v in m._last_idx_maps.items()})
def cmp(a, b):
    ga, gb = res[a][0], res[b][0]
    gmax = max(float(v.abs().max()) for v in gb.values())
    bad = sorted(((float((ga[k] - gb[k]).abs().max() / gb[k].abs().max().clamp_min(1e-4 * gmax)), k) for k in ga), reverse=True)
    for k in ga:
        if "network.4." in k:
            print("   ", k, float((ga[k] - gb[k]).abs().max() / gb[k].abs().max().clamp_min(1e-4 * gmax)), float(gb[k].abs().max()))
    flips = sum(int((res[a][3][k] != res[b][3][k]).sum()) for k in res[a][3])
    print(a, "vs", b, "idx flips", flips, "dx", float((res[a][1] - res[b][1]).abs().max() / res[b][1].abs().max()), "worst params", [(f"{e:.2e}", k) for e, k in bad[:4]], "n>1e-3:", sum(1 for e, _ in bad if e > 1e-3))
cmp("masked", "pair")
