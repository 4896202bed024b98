"""Is there device idle time between consecutive replays of the captured step?  (diagnostic)

    python tools/debug/graph_gap.py [steps=12]

Per replay: host time of the call, device time inside the replay (events around it), device time between the end of one
replay and the start of the next.  Then the same loop alternating TWO captured copies of the step."""
import sys
import time

import torch

sys.path.insert(0, ".")
import asy_vrnet_amd as A      # noqa: E402
from asy_vrnet_amd.graph import GraphedStep      # noqa: E402
from asy_vrnet_amd.losses import mean_square_loss      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda", 0)
    model = A.EfficientVRNet(4, 9, "l", img_size=512).to(dev).train()
    A.randomize_state_dict(model.state_dict(), seed=0)
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((8, 3, 512, 512), generator=g).to(dev)
    r = torch.randn((8, 4, 512, 512), generator=g).to(dev)
    gs = [GraphedStep(model, mean_square_loss, 8, 512, dev)]

    def run(objs, label):
        for i in range(3):
            objs[i % len(objs)](x, r)
        torch.cuda.synchronize()
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        eb = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        host = []
        t0 = time.perf_counter()
        for i in range(steps):
            h0 = time.perf_counter()
            ea[i].record()
            objs[i % len(objs)](x, r)
            eb[i].record()
            host.append(1e3 * (time.perf_counter() - h0))
        torch.cuda.synchronize()
        wall = 1e3 * (time.perf_counter() - t0) / steps
        inside = [ea[i].elapsed_time(eb[i]) for i in range(steps)]
        gaps = [eb[i].elapsed_time(ea[i + 1]) for i in range(steps - 1)]
        print(f"{label}: wall {wall:.3f} ms/step; host call ms {[round(h, 2) for h in host]}")
        print(f"   device ms inside a replay {[round(v, 2) for v in inside]}")
        print(f"   device ms between replays {[round(v, 3) for v in gaps]}")

    run(gs, "one captured copy")
    gs.append(GraphedStep(model, mean_square_loss, 8, 512, dev))
    run(gs, "two captured copies, alternating")
    run(gs[:1], "one captured copy again")


if __name__ == "__main__":
    main()
