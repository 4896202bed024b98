"""Debug: checksums of every forward kernel output (conv2d, affine, cluster_fwd, GN coefficient kernels) per run; on a run whose
forward differs from run 0, print the first records that differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip, program
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 8
net = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(net.state_dict(), seed=2)
net.compute_dtype = dtype
x, r = A.synthetic_inputs(bs, 512, 11, "cuda")
state = {k: v.clone() for k, v in net.state_dict().items()}
rec, on = [], [False]
def ck(t):
    return t.double().sum() if t is not None else None
def wrap(name, outs):
    orig = getattr(hip, name)
    def f(*a, **k):
        orig(*a, **k)
        if on[0] and not k.get("out_nchw") and not k.get("accumulate"):
            st = torch.cuda.current_stream().cuda_stream
            for i in outs:
                t = a[i] if isinstance(i, int) else k.get(i)
                if t is not None:
                    rec.append((name, i, tuple(t.shape), st, ck(t)))
    setattr(hip, name, f)
wrap("conv2d", [4, "ypre"]); wrap("affine", [0]); wrap("cluster_fwd", [5]); wrap("gn_coef_from_pairs", [8, 9, 10]); wrap("gn_stats_fwd", [8, 9, 10])
wrap("bn_stats_fwd", [12, 13, 14]); wrap("patch_gather", [3]); wrap("dwconv3x3", [3])
ref = None
for it in range(runs):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    rec.clear(); on[0] = True
    det, seg = net(x, r)
    on[0] = False
    (seg.mean() + sum(d.mean() for d in det)).backward()
    torch.cuda.synchronize()
    cur = [(n, i, s, st, float(c)) for n, i, s, st, c in rec]
    if ref is None:
        ref = cur
        streams = sorted({c[3] for c in cur})
        print("records per forward:", len(cur), "streams", len(streams))
        continue
    # compare per stream (issue order within a stream is fixed)
    diffs = []
    for st in sorted({c[3] for c in cur}):
        a = [c for c in ref if c[3] == st]; b = [c for c in cur if c[3] == st]
        for j, (p, q) in enumerate(zip(a, b)):
            if p[4] != q[4]:
                diffs.append((st, j, p[0], p[1], p[2], p[4], q[4]))
                break
    if diffs:
        print(f"run {it}: first differing record per stream:")
        for d in diffs:
            print("   ", d)
