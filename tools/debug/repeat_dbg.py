"""Debug: is one forward+backward bit-repeatable?  Lists the parameters whose gradient differs between runs.
    python tools/debug/repeat_dbg.py [dtype] [concurrent 0/1] [runs] [batch] [pair]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 6
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 16
net = A.EfficientVRNet(4, 9, "l", img_size=512).cuda().train()
A.randomize_state_dict(net.state_dict(), seed=2)
net.compute_dtype = dtype
net.concurrent = bool(conc)
net.pair_streams = len(sys.argv) > 5 and sys.argv[5] == "pair"
x, r = A.synthetic_inputs(bs, 512, 11, "cuda")
state = {k: v.clone() for k, v in net.state_dict().items()}
def cks(t):
    t = t.detach().double().flatten()
    return (float(t.sum()), float((t * t).sum()))
base = {"x": cks(x), "r": cks(r)}
base.update({"state." + k: cks(v) for k, v in state.items()})
reported = False
ref = None
for it in range(runs):
    net.load_state_dict(state)
    net.zero_grad(set_to_none=True)
    det, seg = net(x, r)
    fwd_out = [seg.detach().clone()] + [d.detach().clone() for d in det]
    g = torch.Generator(device="cuda").manual_seed(1)
    gd = [torch.randn(d.shape, device="cuda", generator=g) / d.numel() for d in det]
    gs = torch.randn(seg.shape, device="cuda", generator=g) / seg.numel()
    torch.autograd.backward([*det, seg], [*gd, gs])
    torch.cuda.synchronize()
    cur = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
    cur["<seg>"] = seg.detach().clone()
    for i, d in enumerate(det):
        cur[f"<det{i}>"] = d.detach().clone()
    cur["<fwd>"] = torch.cat([t.flatten() for t in fwd_out])
    if ref is None:
        ref = cur
        continue
    bad = [(k, float((cur[k].double() - ref[k].double()).abs().max() / ref[k].double().abs().max().clamp_min(1e-30))) for k in ref
           if not torch.equal(cur[k], ref[k])]
    if it > 0 and not torch.equal(cur["<fwd>"], ref["<fwd>"]):
        print(f"run {it}: FORWARD outputs differ right after the forward")
    if bad or it == runs - 1:
        outs = {k: float((cur[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ("<seg>", "<det0>", "<det1>", "<det2>")}
        print(f"run {it}: {len(bad)} of {len(ref)} tensors differ; output rel diffs: {outs}")
    if bad and len(bad) <= 6:
        print("   differing:", [(k.replace("backbone.backbone.", "bb."), f"{e:.1e}") for k, e in bad])
        for k, _ in bad:
            dd = (cur[k].double() - ref[k].double()).flatten(1) if cur[k].dim() > 1 else (cur[k].double() - ref[k].double())[None]
            nz = (dd != 0)
            print("     ", k, "shape", tuple(cur[k].shape), "differing elements", int(nz.sum()), "per row", nz.sum(1).tolist(),
                  "cols", nz.any(0).nonzero().flatten().tolist()[:40], "ratio", (dd[nz] / ref[k].double().flatten(1)[nz] if cur[k].dim() > 1 else dd[nz])[:6].tolist())
    if bad and not reported:
        reported = True
        print("   differing:", [(k.replace("backbone.backbone.", "bb."), f"{e:.1e}") for k, e in bad][:40])
        now = {"x": cks(x), "r": cks(r)}
        now.update({"state." + k: cks(v) for k, v in state.items()})
        print("   persistent tensors that changed:", [k for k in base if base[k] != now[k]])
        fq = getattr(net, "_fused_qkv", None)
        if fq is not None:
            print("   fused qkv copies stale:", [i for i, (d, s_) in enumerate(zip(fq.dst, fq.src)) if not torch.equal(d, s_.detach().view_as(d))])
        print("   params vs state:", [k for k, v in net.state_dict().items() if not torch.equal(v, state[k])][:20])
    if bad and False:
        names = [k for k, _ in bad]
        same = [k for k in ref if k not in names]
        print("   unchanged:", [k.replace("backbone.backbone.", "bb.") for k in same][:60])
