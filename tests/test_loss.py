"""Training losses: the oracle restatement against the reference's own YOLOLoss / CE / Focal / Dice
(fixture from tools/make_golden_loss.py); the HIP losses against the oracle and the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as LO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_small.npz")


def setup():
    z = np.load(GOLD)
    B, S, NC, NS = (int(v) for v in z["shape"])
    dets, seg, weights = LO.synthetic_preds(B, S, NC, NS, seed=7)
    labels, pngs, seg_labels = LO.synthetic_targets(B, S, NC, NS, seed=3, empty=(1,))
    return z, (B, S, NC, NS), dets, seg, weights, labels, pngs, seg_labels


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def close_fp(got, want, tol, what):
    """fingerprint = [sum, l2, max|.|]; the sum of a softmax gradient is analytically ~0, so it is compared on the
    scale of the l2 norm, not of itself."""
    scale = abs(want[1])
    for a, b in zip(got, want):
        assert abs(a - b) <= tol * scale, (what, got, want)


def test_oracle_yolo_loss_matches_reference():
    z, (B, S, NC, NS), dets, _, _, labels, _, _ = setup()
    assert len(labels[1]) == 0 and all(len(l) > 0 for i, l in enumerate(labels) if i != 1)       # :145-149 exercised
    ins = [d.clone().requires_grad_(True) for d in dets]
    loss, assigns = LO.yolo_loss(ins, labels, NC, return_assignment=True)
    loss.backward()
    assert abs(loss.item() - float(z["yolo_loss"])) <= 2e-6 * abs(float(z["yolo_loss"]))
    for i, t in enumerate(ins):
        assert rel(t.grad, z[f"yolo_grad{i}"]) < 1e-5, i
        assert torch.equal(dets[i], ins[i].detach())                # inputs are not mutated
    assert sum(int(a[0].sum()) for a in assigns) > 0


def test_oracle_seg_losses_match_reference():
    z, (B, S, NC, NS), _, seg, weights, _, pngs, seg_labels = setup()
    assert (pngs == NS).any()                                       # ignore class present
    for name, fn in (("ce", lambda x: LO.ce_loss(x, pngs, weights, NS)),
                     ("focal", lambda x: LO.focal_loss(x, pngs, weights, NS)),
                     ("dice", lambda x: LO.dice_loss(x, seg_labels))):
        x = seg.clone().requires_grad_(True)
        l = fn(x)
        l.backward()
        assert abs(l.item() - float(z[f"{name}_loss"])) <= 2e-6 * abs(float(z[f"{name}_loss"])), name
        assert rel(x.grad[:, :, ::4, ::4], z[f"{name}_grad_sub"]) < 1e-5, name
        close_fp(LO.fingerprint(x.grad), z[f"{name}_grad_fp"], 1e-5, name)
    tot = LO.seg_loss(seg, pngs, seg_labels, weights, NS, focal=True, dice=True)
    assert abs(tot.item() - float(z["focal_loss"]) - float(z["dice_loss"])) < 1e-5


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_yolo_loss_matches_oracle_and_reference():
    from asy_vrnet_amd import losses
    z, (B, S, NC, NS), dets, _, _, labels, _, _ = setup()
    yl = losses.YOLOLoss(NC).cuda()
    ins = [d.clone().cuda().requires_grad_(True) for d in dets]
    loss = yl(ins, labels)
    (loss * 1.0).backward()
    assert abs(loss.item() - float(z["yolo_loss"])) <= 1e-5 * abs(float(z["yolo_loss"]))
    for i, t in enumerate(ins):
        assert rel(t.grad.cpu(), z[f"yolo_grad{i}"]) < 2e-5, i
        assert torch.equal(t.detach().cpu(), dets[i])               # not mutated (the reference's :108-110 is)
    # the assignment itself, anchor by anchor, against the oracle's SimOTA
    _, assigns = LO.yolo_loss([d.clone() for d in dets], labels, NC, return_assignment=True)
    fg, mg, pi, stats = yl.assignments([d.cuda() for d in dets], labels)
    for b, (ofg, omatched, opious) in enumerate(assigns):
        assert torch.equal(fg[b].cpu(), ofg)
        assert torch.equal(mg[b].cpu()[ofg].long(), omatched)
        assert torch.allclose(pi[b].cpu()[ofg], opious, rtol=1e-5, atol=1e-7)
        assert (mg[b].cpu()[~ofg] == -1).all()
    assert int(stats[1].item()) == sum(int(a[0].sum()) for a in assigns)
    # an all-empty batch: only the objectness term, num_fg clamps to 1 (:175)
    empty = [torch.zeros(0, 5) for _ in range(B)]
    l0 = yl([d.cuda() for d in dets], empty)
    assert abs(l0.item() - LO.yolo_loss(dets, empty, NC).item()) <= 1e-5 * abs(l0.item())
    # scaled upstream gradient
    ins2 = [d.clone().cuda().requires_grad_(True) for d in dets]
    (3.0 * yl(ins2, labels)).backward()
    assert rel(ins2[0].grad.cpu(), 3.0 * z["yolo_grad0"]) < 2e-5


@pytest.mark.gpu
def test_hip_yolo_loss_many_boxes_and_full_size():
    """512 px (5376 anchors), bs 4, up to 40 boxes per image incl. one empty image: against the oracle."""
    from asy_vrnet_amd import losses
    B, S, NC, NS = 4, 512, 4, 9
    dets, _, _ = LO.synthetic_preds(B, S, NC, NS, seed=21)
    rng = np.random.default_rng(9)
    labels = []
    for b in range(B):
        n = 0 if b == 2 else int(rng.integers(20, 41))
        box = np.concatenate([rng.uniform(40, 470, (n, 2)), rng.uniform(12, 260, (n, 2)), rng.integers(0, NC, (n, 1))], 1)
        labels.append(torch.from_numpy(box.astype(np.float32)).reshape(n, 5))
    ref_in = [d.clone().requires_grad_(True) for d in dets]
    ref = LO.yolo_loss(ref_in, labels, NC)
    ref.backward()
    ins = [d.clone().cuda().requires_grad_(True) for d in dets]
    loss = losses.YOLOLoss(NC).cuda()(ins, labels)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item())
    for a, b in zip(ins, ref_in):
        assert rel(a.grad.cpu(), b.grad) < 5e-5


@pytest.mark.gpu
def test_hip_seg_losses_match_oracle_and_reference():
    from asy_vrnet_amd import losses
    z, (B, S, NC, NS), _, seg, weights, _, pngs, seg_labels = setup()
    for name, fn in (("ce", lambda x: losses.CE_Loss(x, pngs.cuda(), weights.cuda(), num_classes=NS)),
                     ("focal", lambda x: losses.Focal_Loss(x, pngs.cuda(), weights.cuda(), num_classes=NS)),
                     ("dice", lambda x: losses.Dice_loss(x, seg_labels.cuda()))):
        x = seg.clone().cuda().requires_grad_(True)
        l = fn(x)
        l.backward()
        assert abs(l.item() - float(z[f"{name}_loss"])) <= 1e-5 * abs(float(z[f"{name}_loss"])), name
        assert rel(x.grad[:, :, ::4, ::4].cpu(), z[f"{name}_grad_sub"]) < 3e-5, name
        close_fp(LO.fingerprint(x.grad.cpu()), z[f"{name}_grad_fp"], 3e-5, name)
    # CE without class weights, and the combination of utils_fit.py:96-106 with its factor 5
    x = seg.clone().cuda().requires_grad_(True)
    l = losses.CE_Loss(x, pngs.cuda(), None, num_classes=NS)
    xr = seg.clone().requires_grad_(True)
    lr_ = LO.ce_loss(xr, pngs, None, NS)
    assert abs(l.item() - lr_.item()) < 1e-5
    dets, _, _ = LO.synthetic_preds(B, S, NC, NS, seed=7)
    labels, _, _ = LO.synthetic_targets(B, S, NC, NS, seed=3, empty=(1,))
    ins = [d.clone().cuda().requires_grad_(True) for d in dets]
    xs = seg.clone().cuda().requires_grad_(True)
    total, ldet, lseg = losses.training_loss(losses.YOLOLoss(NC).cuda(), ins, xs, labels, pngs.cuda(), seg_labels.cuda(),
                                             weights.cuda(), NS, focal_loss=True, dice_loss=True)
    total.backward()
    want = float(z["yolo_loss"]) + 5 * (float(z["focal_loss"]) + float(z["dice_loss"]))
    assert abs(total.item() - want) <= 1e-5 * want
    assert abs(lseg.item() - float(z["focal_loss"]) - float(z["dice_loss"])) < 1e-5 and abs(ldet.item() - float(z["yolo_loss"])) < 1e-3
    rs = seg.clone().requires_grad_(True)
    (5 * LO.seg_loss(rs, pngs, seg_labels, weights, NS, True, True)).backward()
    assert rel(xs.grad.cpu(), rs.grad) < 3e-5
    assert rel(ins[1].grad.cpu(), z["yolo_grad1"]) < 2e-5
    with pytest.raises(RuntimeError):
        losses.CE_Loss(seg[:, :, ::2, ::2].contiguous().cuda(), pngs.cuda(), None, num_classes=NS)


@pytest.mark.gpu
def test_mean_square_driver_matches_the_torch_expression():
    """losses.mean_square_loss -- the synthetic scalar behind bench.py's backward pass -- against the eager torch
    expression it replaces: value 1e-6, gradients 1e-6 (also with an upstream gradient other than 1), repeatable bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from asy_vrnet_amd.losses import mean_square_loss
    g = torch.Generator().manual_seed(3)
    shapes = [(2, 9, 64, 64), (2, 9, 32, 32), (2, 9, 16, 16), (2, 9, 512, 512)]
    ts = [torch.randn(s, generator=g).cuda().requires_grad_() for s in shapes]
    ref = [t.detach().clone().requires_grad_() for t in ts]
    want = sum((d * d).mean() for d in ref[:3]) + (ref[3] * ref[3]).mean()
    (want * 0.7).backward()
    got = mean_square_loss(ts[:3], ts[3])
    (got * 0.7).backward()
    assert abs(float(got.detach()) - float(want.detach())) < 1e-6 * abs(float(want.detach()))
    for a, b in zip(ts, ref):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-12)
    again = mean_square_loss([t.detach() for t in ts[:3]], ts[3].detach())
    assert torch.equal(again, got.detach())
