"""Bitwise repeatability of the REPLAYED step: N replays of one captured step on the same batch, every parameter gradient and the
loss compared with the first replay.   python tools/debug/replay_soak.py [replays=300] [bf16]"""
import sys

import torch

sys.path.insert(0, ".")
import asy_vrnet_amd as A      # noqa: E402
from asy_vrnet_amd.graph import GraphedStep      # noqa: E402
from asy_vrnet_amd.losses import mean_square_loss      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    bf16 = "bf16" in sys.argv
    dev = torch.device("cuda", 0)
    model = A.EfficientVRNet(4, 9, "l", img_size=512).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    if bf16:
        model.compute_dtype = "bf16"
    bs = 16 if bf16 else 8
    x, r = A.synthetic_inputs(bs, 512, 5, dev)
    gs = GraphedStep(model, mean_square_loss, bs, 512, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ref, bad = None, 0
    for i in range(n):
        model.load_state_dict(state)          # BatchNorm running statistics back to the same start
        loss = gs(x, r)
        torch.cuda.synchronize()
        cur = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None] + [loss.flatten()])
        if ref is None:
            ref = cur.clone()
        elif not torch.equal(cur, ref):
            bad += 1
            print(f"replay {i}: {int((cur != ref).sum())} of {cur.numel()} values differ")
    print(f"{n} replays ({'bf16 bs 16' if bf16 else 'fp32 bs 8'}): {bad} differed from the first")


if __name__ == "__main__":
    main()
