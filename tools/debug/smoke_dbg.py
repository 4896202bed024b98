import sys, os
sys.path.insert(0, os.getcwd())
import torch
import asy_vrnet_amd as A
from tests.parity import compare_with_oracle
for batch in (2, 4):
    for iseed in (1, 2, 3, 4, 5, 6):
        torch.manual_seed(0)
        model = A.EfficientVRNet(4, 9, "nano", img_size=128).cuda().train()
        A.randomize_state_dict(model.state_dict(), seed=11)
        rep = compare_with_oracle(model, batch=batch, size=128, iseed=iseed, check_grads=True, oracle_dtype=torch.float64)
        print("batch", batch, "iseed", iseed, "grad_err %.2e" % rep["grad_err"], rep["grad_worst"], "ref32 %.2e" % rep["ref32_grad_err"], rep["ok"])
