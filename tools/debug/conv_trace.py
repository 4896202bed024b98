"""Diagnostic: checksum of the output of every conv2d launch of one forward+backward (compare two runs, e.g. with and
without VRNET_IGEMM_DMA=0, with diff): python tools/conv_trace.py phi size batch > trace.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip
from oracle import vrnet_oracle as O

phi, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=21)
m.concurrent = False
x, r = A.synthetic_inputs(batch, size, 31, "cuda")
x.requires_grad_(True); r.requires_grad_(True)
orig = hip.conv2d
n = [0]
saved = []


def conv2d(*a, **k):
    orig(*a, **k)
    out = a[4]
    B, H, W, Ci, OH, OW, Co, kh, kw, s, p, d = a[6:18]
    flags = {kk: (vv is not None if not isinstance(vv, (int, float)) else vv) for kk, vv in k.items() if kk not in ("stats",)}
    o = out.double()
    print(n[0], "mode", k.get("mode", 0), (B, H, W, Ci, OH, OW, Co, kh, s, d), "lda", a[1], "ldo", a[5], flags,
          f"{float(o.sum()):.10e} {float(o.abs().sum()):.10e}")
    if k.get("mode", 0) == 1 and kh == 1 and s == 1 and not k.get("accumulate") and k.get("kscale") is None and k.get("aux") is None \
            and a[1] == Co and Ci > 4:
        rows = B * H * W
        av = a[0].reshape(-1)[:rows * a[1]].reshape(rows, a[1]).double()       # dY [M][Cout]
        wv = a[2].reshape(Co, Ci).double()
        ref = av @ wv
        got = out.reshape(-1)[:rows * a[5]].reshape(rows, a[5])[:, :Ci].double()
        d = (ref - got).abs()
        e = float(d.max() / ref.abs().max())
        if e > 1e-4:
            badrows = torch.nonzero(d.max(dim=1).values > 1e-4 * ref.abs().max()).flatten().tolist()
            r0 = badrows[0]
            print("   !! call", n[0], "vs fp64 matmul: err", e, "bad rows", badrows[:12], "finite:", bool(torch.isfinite(av).all()),
                  "row", r0, "dY row absmax", float(av[r0].abs().max()), "next row absmax", float(av[min(r0 + 1, rows - 1)].abs().max()),
                  "got/ref", got[r0, :4].tolist(), ref[r0, :4].tolist())
    n[0] += 1
    if os.environ.get('CT_SAVE'):
        saved.append(out.detach().cpu().clone())


hip.conv2d = conv2d
orig_affine = hip.affine
masks = []


def affine(out, *a, **k):
    orig_affine(out, *a, **k)
    if k.get("pre", 0) == 1 and os.environ.get("CT_SAVE"):
        masks.append((out > 0).cpu())


hip.affine = affine
import asy_vrnet_amd.program as P
det, seg = m(x, r)
O.synthetic_loss(det, seg).backward()
torch.cuda.synchronize()
if os.environ.get('CT_SAVE'):
    torch.save(saved, os.environ['CT_SAVE'])
    torch.save(masks, os.environ['CT_SAVE'] + '.masks')
