"""HIP path vs the reference's own 512 px vectors (fp32 and fp64 fixtures): error quantiles (diagnostic)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import asy_vrnet_amd as A
for name in ("net_nano_512_train", "net_nano_512_train_fp64"):
    z = np.load(os.path.join(ROOT, "tests/golden", name + ".npz"))
    meta = json.load(open(os.path.join(ROOT, "tests/golden", name + ".json")))
    m = A.EfficientVRNet(4, 9, meta["phi"], img_size=meta["size"]).cuda().train()
    A.randomize_state_dict(m.state_dict(), seed=meta["pseed"])
    x, r = A.synthetic_inputs(meta["batch"], meta["size"], meta["iseed"])
    xg, rg = x.cuda().requires_grad_(True), r.cuda().requires_grad_(True)
    det, seg = m(xg, rg)
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
    s = meta["seg_stride"]
    for k, a in (("det0", det[0]), ("det1", det[1]), ("det2", det[2]), ("seg", seg[:, :, ::s, ::s]), ("dx", xg.grad[:, :, ::s, ::s]), ("dr", rg.grad[:, :, ::s, ::s])):
        b = torch.from_numpy(z[k]).double(); a = a.detach().double().cpu()
        e = (a - b).abs() / b.abs().max()
        print(name, k, "max", float(e.max()), "q99", float(e.flatten().quantile(0.99)), "q999", float(e.flatten().kthvalue(int(0.999 * e.numel()))[0]), "frac>1e-3", float((e > 1e-3).double().mean()))
    pd = dict(m.named_parameters())
    worst = []
    for k in z.files:
        if k.startswith("g:"):
            b = torch.from_numpy(z[k]).double(); a = pd[k[2:]].grad.double().cpu()
            worst.append((float((a - b).abs().max() / b.abs().max().clamp_min(1e-12)), k))
    print(name, "grads worst", sorted(worst, reverse=True)[:5])
