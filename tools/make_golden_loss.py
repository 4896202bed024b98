"""Golden vectors for the training losses from the REFERENCE's own classes, run in the build
container: YOLOLoss (nets/yolo_training.py), CE_Loss / Focal_Loss / Dice_loss (nets/deeplabv3_training.py).

    python tools/make_golden_loss.py         # writes tests/golden/loss_small.npz

Inputs (det maps, seg logits, class weights) and targets are regenerated from seeds by
oracle.loss_oracle.synthetic_preds / synthetic_targets; the fixture holds the reference's losses and gradients
(seg gradients: every 4th pixel plus a fingerprint of the whole tensor).  The reference's YOLOLoss mutates its inputs in place, so it gets clones."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_harness import load_reference  # noqa: E402
from oracle import loss_oracle as LO  # noqa: E402

load_reference()
from nets.yolo_training import YOLOLoss  # noqa: E402
from nets.deeplabv3_training import CE_Loss, Dice_loss, Focal_Loss  # noqa: E402

if __name__ == "__main__":
    B, S, NC, NS = 3, 128, 4, 9
    dets, seg, weights = LO.synthetic_preds(B, S, NC, NS, seed=7)
    labels, pngs, seg_labels = LO.synthetic_targets(B, S, NC, NS, seed=3, empty=(1,))
    out = {"shape": np.array([B, S, NC, NS])}

    yl = YOLOLoss(NC, fp16=False)
    ins = [d.clone().requires_grad_(True) for d in dets]
    loss = yl([i * 1.0 for i in ins], labels)            # * 1.0: the in-place decode needs a non-leaf
    loss.backward()
    out["yolo_loss"] = loss.detach().numpy()
    for i, t in enumerate(ins):
        out[f"yolo_grad{i}"] = t.grad.numpy()
    for name, fn in (("ce", lambda x: CE_Loss(x, pngs, weights, num_classes=NS)),
                     ("focal", lambda x: Focal_Loss(x, pngs, weights, num_classes=NS)),
                     ("dice", lambda x: Dice_loss(x, seg_labels))):
        x = seg.clone().requires_grad_(True)
        l = fn(x)
        l.backward()
        out[f"{name}_loss"] = l.detach().numpy()
        out[f"{name}_grad_sub"] = x.grad[:, :, ::4, ::4].numpy()          # every 4th pixel + a fingerprint of all
        out[f"{name}_grad_fp"] = np.array(LO.fingerprint(x.grad))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss_small.npz"), **out)
    print("wrote loss_small.npz", {k: float(v) for k, v in out.items() if k.endswith("_loss")})
