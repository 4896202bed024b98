"""Which weight-plane entries are (re)registered in the 2nd / 3rd forward (diagnostic)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from asy_vrnet_amd import hip, program
step = [0]
log = collections.Counter()
orig = program.WeightPlanes.get
def get(self, key, w, J, K, sj, sk, kscale):
    ent = self.entries.get(key)
    ids = (w.data_ptr(), None if kscale is None else kscale.data_ptr())
    if step[0] >= 1 and not (ent is not None and ent[7] == ids):
        why = "new key" if ent is None else ("w moved" if ent[7][0] != ids[0] else "kscale moved")
        site = [f"{fr.name}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if fr.filename.endswith("program.py")][-2:]
        log[(step[0], why, J, K, tuple(site))] += 1
    return orig(self, key, w, J, K, sj, sk, kscale)
program.WeightPlanes.get = get
packs = collections.Counter()
for n in ("conv_planes_pack", "pack_weight_t", "planes_split", "mlp_pack"):
    if hasattr(hip, n):
        f = getattr(hip, n)
        def mk(n, f):
            def w(*a, **k):
                site = [f"{fr.name}:{fr.lineno}" for fr in traceback.extract_stack()[:-1] if fr.filename.endswith("program.py")][-2:]
                packs[(step[0], n, tuple(site))] += 1
                return f(*a, **k)
            return w
        setattr(hip, n, mk(n, f))
torch.manual_seed(0)
m = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=0)
x, r = torch.randn(8, 3, 512, 512, device="cuda"), torch.randn(8, 4, 512, 512, device="cuda")
for it in range(3):
    step[0] = it
    m.zero_grad(set_to_none=True)
    det, seg = m(x, r)
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
torch.cuda.synchronize()
for k, c in sorted(log.items()):
    print("REG", k, c)
for k, c in sorted(packs.items()):
    if k[0] >= 1:
        print("PACK", k, c)
