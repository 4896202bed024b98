"""Section timeline of the REPLAYED step without a tracer: device-clock stamps at the forks / chain ends / joins of the
backbone sections (model.debug_stamps -> program.RT.stamp -> vrnet_clock_stamp).

    python tools/debug/section_stamps.py [--bf16] [--batch N] [--size S]"""
import sys

import torch

sys.path.insert(0, ".")
import asy_vrnet_amd as A      # noqa: E402
from asy_vrnet_amd.graph import GraphedStep      # noqa: E402
from asy_vrnet_amd.losses import mean_square_loss      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = A.EfficientVRNet(4, 9, "l", img_size=512 if "--size" not in sys.argv else int(sys.argv[sys.argv.index("--size") + 1])).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    model.debug_stamps = True
    if "--bf16" in sys.argv:
        model.compute_dtype = "bf16"
    bs = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 8
    size = int(sys.argv[sys.argv.index("--size") + 1]) if "--size" in sys.argv else 512
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((bs, 3, size, size), generator=g).to(dev)
    r = torch.randn((bs, 4, size, size), generator=g).to(dev)
    gs = GraphedStep(model, mean_square_loss, bs, size, dev)
    for _ in range(6):
        gs(x, r)
    torch.cuda.synchronize()
    names = model._stamp_names
    t = model._stamp_buf[:len(names)].cpu().tolist()
    t0 = t[0]
    for n, v in zip(names, t):
        print(f"{(v - t0) / 100.0:9.1f} us  {n}")


if __name__ == "__main__":
    main()
