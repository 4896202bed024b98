"""Which parameter gradients differ between model.plane_gemms = <kinds> and the in-kernel-split path (teacher-forced)."""
import sys
import torch
sys.path.insert(0, ".")
import asy_vrnet_amd as A

phi, bs = sys.argv[1], int(sys.argv[2])
net = A.EfficientVRNet(4, 9, phi, img_size=512).cuda().train()
A.randomize_state_dict(net.state_dict(), seed=2)
x, r = A.synthetic_inputs(bs, 512, 5, "cuda")
bn = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}


def restore():
    sd = net.state_dict()
    for k, v in bn.items():
        sd[k].copy_(v)


def run(gdet, gseg):
    net.zero_grad(set_to_none=True)
    net.record_relu_masks = True
    det, seg = net(x, r)
    net.record_relu_masks = False
    run.masks = {k: v.clone() for k, v in net._last_relu_masks.items()}
    torch.autograd.backward([*det, seg], [*gdet, gseg])
    return [d.detach().clone() for d in det], seg.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


net.plane_gemms = False
with torch.no_grad():
    det, seg = net(x, r)
g = torch.Generator(device="cuda").manual_seed(3)
gdet = [torch.randn(d.shape, device="cuda", generator=g) / d.numel() for d in det]
gseg = torch.randn(seg.shape, device="cuda", generator=g) / seg.numel()
restore()
d0, s0, g0 = run(gdet, gseg)
m0 = run.masks
net.forced_idx_maps = {k: v.clone() for k, v in net._last_idx_maps.items()}
restore()
d0b, s0b, g0b = run(gdet, gseg)          # the same path again, teacher-forced: the noise floor (must be 0)
print("same path twice: max grad diff", max(float((g0b[k] - g0[k]).abs().max()) for k in g0))
for kinds in sys.argv[3:]:
    restore()
    net.plane_gemms = kinds
    d1, s1, g1 = run(gdet, gseg)
    flips = {k: int((run.masks[k] != m0[k]).sum()) for k in m0}
    print(kinds, "ReLU mask bits decided differently:", sum(flips.values()), "of", sum(v.numel() for v in m0.values()), {k: v for k, v in flips.items() if v})
    errs = sorted(((float((g1[k].double() - g0[k].double()).norm() / g0[k].double().norm().clamp_min(1e-30)), k) for k in g0), reverse=True)
    num = sum(float(((g1[k].double() - g0[k].double()) ** 2).sum()) for k in g0)
    den = sum(float((g0[k].double() ** 2).sum()) for k in g0)
    print(kinds, "aggregate", (num / den) ** 0.5, "outputs", [float((a - b).abs().max() / b.abs().max()) for a, b in zip(d1 + [s1], d0 + [s0])])
    for e, k in errs[:12]:
        print("   %.3e  %s  |g| %.3e" % (e, k, float(g0[k].norm())))
