import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asy_vrnet_amd as A
from tests.parity import compare_with_oracle
phi, size, batch, pseed, iseed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
m = A.EfficientVRNet(4, 9, phi, img_size=size).cuda().train()
A.randomize_state_dict(m.state_dict(), seed=pseed)
if os.environ.get("PAIR"):
    m.pair_streams = True
if os.environ.get("SERIAL"):
    m.concurrent = False
rep = compare_with_oracle(m, batch, size, iseed=iseed, check_grads=True, oracle_dtype=torch.float64, per_param=True)
pp = rep.pop("_per_param")
print({k: v for k, v in rep.items()})
bad = [(k, e) for k, e in pp if e > 2e-3]
print(len(bad), "of", len(pp), "parameters above 2e-3; in forward order:")
for k, e in bad[:60]:
    print(f"  {e:9.2e}  {k}")
