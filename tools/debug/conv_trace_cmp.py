"""Diagnostic: compares two CT_SAVE dumps of tools/conv_trace.py call by call."""
import sys
import torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
lines = [l for l in open(sys.argv[3]).read().splitlines() if l[:1].isdigit()]
for i, (x, y) in enumerate(zip(a, b)):
    d = (x.double() - y.double()).abs()
    e = float(d.max() / y.double().abs().max().clamp_min(1e-30))
    if e > 1e-4:
        idx = int(d.reshape(-1).argmax())
        print(f"{e:.3e} bad={int((d > 1e-4 * y.abs().max()).sum())}/{d.numel()} argmax={idx} (row {idx // x.shape[-1]}, col {idx % x.shape[-1]}) {lines[i][:170]}")

import os
if os.path.exists(sys.argv[1] + ".masks"):
    ma, mb = torch.load(sys.argv[1] + ".masks"), torch.load(sys.argv[2] + ".masks")
    for i, (x, y) in enumerate(zip(ma, mb)):
        d = int((x != y).sum())
        if d:
            idx = torch.nonzero((x != y).reshape(-1, x.shape[-1]))
            print(f"relu mask {i} shape {tuple(x.shape)}: {d} elements differ at (row, col) {idx[:4].tolist()}")
