"""Diagnostic: whole-net parity report for one (phi, size, batch, pseed, iseed)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asy_vrnet_amd as A
from tests.parity import compare_with_oracle

phi, size, batch, pseed, iseed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=pseed)
if os.environ.get('VRNET_SERIAL') == '1':
    m.concurrent = False
rep = compare_with_oracle(m, batch, size, iseed=iseed, check_grads=True, oracle_dtype=torch.float64, per_param=True)
per = rep.pop("_per_param", [])
for k, e in reversed(per):
    if e > 1e-3:
        print(f"  {e:.3e} {k}")
print({k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in rep.items()})
