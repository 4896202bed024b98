"""Which program lines launch the small chain kernels (add_, copy_channels, cat2, affine, moments ...) in one training step
(diagnostic): counts per (wrapper, program.py line)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip

names = sys.argv[1:] or ["add_", "copy_channels", "cat2", "affine", "moments", "fill_", "upsample", "upsample_bwd", "gn_apply_fwd", "gn_apply_bwd", "bn_stats_bwd_zmask", "bn_apply_bwd_zmask", "bn_coef_fwd_from_partials"]
counts = collections.Counter()
def wrap(name):
    fn = getattr(hip, name)
    def w(*a, **k):
        for fr in reversed(traceback.extract_stack()[:-1]):
            if fr.filename.endswith("program.py"):
                counts[(name, fr.lineno, fr.name)] += 1
                break
        return fn(*a, **k)
    setattr(hip, name, w)
for n in names:
    if hasattr(hip, n):
        wrap(n)
torch.manual_seed(0)
m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=0)
x, r = torch.randn(8, 3, 512, 512, device="cuda"), torch.randn(8, 4, 512, 512, device="cuda")
for it in range(2):
    counts.clear()
    m.zero_grad(set_to_none=True)
    det, seg = m(x, r)
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
torch.cuda.synchronize()
tot = collections.Counter()
for (n, ln, fn), c in sorted(counts.items(), key=lambda kv: (kv[0][0], -kv[1])):
    tot[n] += c
    print(f"{n:28s} program.py:{ln:5d} {fn:28s} {c:4d}")
print(dict(tot))
