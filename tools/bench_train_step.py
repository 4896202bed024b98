"""Times the pieces of one REAL training step around the hot path on one GPU:
forward+backward with the reference's loss (YOLOLoss SimOTA + focal + dice, total = det + 5 seg) instead of bench.py's
synthetic scalar, the fused SGD step and the fused EMA update.  Eager launches, HIP events per piece.

    python tools/bench_train_step.py [--phi l] [--batch 8] [--size 512] [--steps 5]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import asy_vrnet_amd as A
from asy_vrnet_amd import losses, optim


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phi", default="l")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, S, NC, NS = a.batch, a.size, 4, 9
    model = A.EfficientVRNet(NC, NS, a.phi, img_size=(S, S)).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    yl = losses.YOLOLoss(NC).to(dev)
    opt = optim.build_optimizer(model, "sgd", 1.25e-3, 0.937, 5e-4)
    ema = optim.ModelEMA(model)
    rng = np.random.default_rng(0)
    x, r = A.synthetic_inputs(B, S, 1, dev)
    labels = [torch.from_numpy(np.concatenate([rng.uniform(60, S - 60, (n, 2)), rng.uniform(16, 200, (n, 2)),
                                               rng.integers(0, NC, (n, 1))], 1).astype(np.float32))
              for n in rng.integers(3, 25, B)]
    png = torch.from_numpy(np.kron(rng.integers(0, NS + 1, (B, S // 16, S // 16)), np.ones((16, 16), dtype=np.int64))).to(dev)
    onehot = torch.nn.functional.one_hot(png, NS + 1).float()
    weights = torch.ones(NS, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    names = ["forward", "loss (value + head grads)", "backward", "sgd step", "ema update"]
    tot = dict.fromkeys(names, 0.0)
    for it in range(a.steps + 2):
        marks = [ev() for _ in range(6)]
        marks[0].record()
        det, seg = model(x, r)
        marks[1].record()
        total, ldet, lseg = losses.training_loss(yl, det, seg, labels, png, onehot, weights, NS, True, True)
        marks[2].record()
        total.backward()
        marks[3].record()
        opt.step()
        marks[4].record()
        ema.update(model)
        marks[5].record()
        opt.zero_grad()
        torch.cuda.synchronize()
        if it >= 2:
            for i, n in enumerate(names):
                tot[n] += marks[i].elapsed_time(marks[i + 1])
    print(f"phi={a.phi} bs={B} {S}x{S}: loss_det={ldet.item():.4f} loss_seg={lseg.item():.4f} (eager, ms per step)")
    for n in names:
        print(f"  {n:28s} {tot[n] / a.steps:8.3f}")
    print(f"  {'sum':28s} {sum(tot.values()) / a.steps:8.3f}")


if __name__ == "__main__":
    main()
