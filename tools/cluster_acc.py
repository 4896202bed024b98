"""Accuracy of the Cluster kernels against the fp64 oracle core (forced assignment): max relative errors."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from asy_vrnet_amd import hip
from oracle import vrnet_oracle as O

def rnd(*shape, seed=0):
    return torch.from_numpy(np.random.default_rng([seed, len(shape)] + list(shape)).standard_normal(shape).astype(np.float32))

for case in [(4, 4, 24, 8, 8, 2), (4, 4, 24, 16, 16, 2), (2, 4, 32, 4, 4, 1), (2, 4, 32, 8, 8, 2), (2, 4, 24, 4, 4, 2), (2, 4, 32, 32, 32, 2), (2, 8, 32, 16, 16, 1)]:
    B, E, D, H, W, fold = case
    f, v, g = rnd(B, E * D, H, W, seed=1), rnd(B, E * D, H, W, seed=2), rnd(B, E * D, H, W, seed=3)
    alpha, beta = torch.tensor([1.3]), torch.tensor([-0.2])
    nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    fg, vg, gg = nh(f), nh(v), nh(g)
    out = torch.empty(B, H, W, E * D, device="cuda"); idx = torch.empty(B, H, W, E, dtype=torch.uint8, device="cuda")
    wgt = torch.empty(B, H, W, E, device="cuda")
    hip.cluster_fwd(fg, vg, E * D, alpha.cuda(), beta.cuda(), out, E * D, idx, wgt, B, H, W, E, D, fold)
    fd, vd = f.double().requires_grad_(True), v.double().requires_grad_(True)
    ad, bd = alpha.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref, _ = O.cluster_core(fd, vd, ad, bd, E, fold, forced_idx=idx.permute(0, 3, 1, 2).contiguous().cpu().long(), report={})
    ref.backward(g.double())
    df, dv = torch.empty_like(fg), torch.empty_like(vg); dab = torch.zeros(2, device="cuda")
    hip.cluster_bwd(fg, vg, E * D, alpha.cuda(), beta.cuda(), idx, gg, E * D, df, dv, E * D, dab[0:1], dab[1:2], 0, B, H, W, E, D, fold)
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    nc = lambda t: t.permute(0, 3, 1, 2)
    print(case, "fwd %.2e df %.2e dv %.2e dalpha %.2e dbeta %.2e" % (rel(nc(out), ref.detach()), rel(nc(df), fd.grad), rel(nc(dv), vd.grad),
          abs(dab[0].item() - ad.grad.item()) / abs(ad.grad.item()), abs(dab[1].item() - bd.grad.item()) / abs(bd.grad.item())))
