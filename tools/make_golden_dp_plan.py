"""Writes tests/golden/dp_ready_pos_<phi>.json: for every parameter, the index of the top-level backward section that issues
(and joins) its last gradient kernel -- what parallel.GradBucketer records in its first backward pass and lays the gradient
arena out by.  Needs a GPU (the recording pass IS a backward pass); the CPU test
tests/test_data_parallel_gloo.py::test_n8_plan_at_l builds the N = 8 segment / bucket plan from it, and the GPU test
tests/test_net_parity.py::test_recorded_order_fixture_is_current keeps it from going stale.
usage (on the GPU box): python tools/make_golden_dp_plan.py [phi ...]  -> gpurun_out/dp_ready_pos_<phi>.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def recorded_positions(phi, size=128, batch=2):
    import asy_vrnet_amd as A
    from asy_vrnet_amd.parallel import DataParallelVRNet
    m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
    A.randomize_state_dict(m.state_dict(), seed=3)
    dp = DataParallelVRNet(m, bucket_bytes=32 << 20)
    x, r = A.synthetic_inputs(batch, size, 1)
    det, seg = dp(x.cuda(), r.cuda())
    (sum((d * d).mean() for d in det) + (seg * seg).mean()).backward()
    torch.cuda.synchronize()
    names = {p: k for k, p in m.named_parameters()}
    bk = dp.bucketer
    return {names[p]: [int(pos), int(bk._rec_order.get(p, -1))] for p, pos in bk.ready_pos.items()}


if __name__ == "__main__":
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    for phi in (sys.argv[1:] or ["l", "nano"]):
        pos = recorded_positions(phi)
        json.dump(pos, open(os.path.join(out, f"dp_ready_pos_{phi}.json"), "w"), indent=0, sort_keys=True)
        print(phi, len(pos), "parameters, positions", min(v[0] for v in pos.values()), "..", max(v[0] for v in pos.values()))
