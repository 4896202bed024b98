"""Does the priority of the stream a captured step is replayed on change the step time?  (diagnostic)

    python tools/debug/launch_priority.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import asy_vrnet_amd as A      # noqa: E402
from asy_vrnet_amd.graph import GraphedStep      # noqa: E402
from asy_vrnet_amd.losses import mean_square_loss      # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model = A.EfficientVRNet(4, 9, "l", img_size=512).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((8, 3, 512, 512), generator=g).to(dev)
    r = torch.randn((8, 4, 512, 512), generator=g).to(dev)
    gs = GraphedStep(model, mean_square_loss, 8, 512, dev)
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    print("priority range", lo, hi)
    streams = {"default stream": None, "normal-priority stream": torch.cuda.Stream(dev, priority=0),
               "high-priority stream": torch.cuda.Stream(dev, priority=-1)}
    for rep in range(2):
        for name, st in streams.items():
            ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
            with ctx:
                for _ in range(3):
                    gs(x, r)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    gs(x, r)
                torch.cuda.synchronize()
                print(f"{name}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step")


if __name__ == "__main__":
    main()
