"""Golden vectors for the training-step updates, produced by the REFERENCE in the build container:
the reference's own parameter grouping code (train.py:460-467, re-run verbatim on the reference model),
torch.optim.SGD / Adam built as train.py:468-473 does, the reference's ModelEMA and get_lr_scheduler.

    python tools/make_golden_optim.py        # writes tests/golden/optim_*.json

Only fingerprints leave this script ([sum, l2, 3 samples] per tensor): parameters and gradients are reproduced
from seeds on both sides (asy_vrnet_amd.randomize_state_dict, oracle.optim_oracle.seeded_grads)."""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_harness import build_reference_model, load_reference  # noqa: E402
import asy_vrnet_amd as A  # noqa: E402
from oracle import optim_oracle as OO  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
load_reference()
from nets.yolo_training import ModelEMA, get_lr_scheduler  # noqa: E402  (reference modules)


def fresh(phi):
    m = build_reference_model(4, 9, phi, img_size=64)
    A.randomize_state_dict(m.state_dict(), seed=11)
    return m


def reference_groups(model):            # train.py:460-467
    pg0, pg1, pg2 = [], [], []
    for k, v in model.named_modules():
        if hasattr(v, "bias") and isinstance(v.bias, nn.Parameter):
            pg2.append(v.bias)
        if isinstance(v, nn.BatchNorm2d) or "bn" in k:
            pg0.append(v.weight)
        elif hasattr(v, "weight") and isinstance(v.weight, nn.Parameter):
            pg1.append(v.weight)
    return pg0, pg1, pg2


def run(phi, optimizer_type, steps=2):
    model = fresh(phi)
    pg0, pg1, pg2 = reference_groups(model)
    ids = {id(p): g for g, ps in enumerate((pg0, pg1, pg2)) for p in ps}
    groups = {n: ids.get(id(p), -1) for n, p in model.named_parameters()}
    lr, momentum, wd = (1e-3, 0.937, 0.0) if optimizer_type == "adam" else (1e-2, 0.937, 5e-4)
    opt = {"adam": lambda: optim.Adam(pg0, lr, betas=(momentum, 0.999)),
           "sgd": lambda: optim.SGD(pg0, lr, momentum=momentum, nesterov=True)}[optimizer_type]()
    opt.add_param_group({"params": pg1, "weight_decay": wd})
    opt.add_param_group({"params": pg2})
    ema = ModelEMA(model)
    ema.updates = 3000                      # mid-training decay (0.777...), so both terms of the average matter
    after = []
    for s in range(steps):
        for n, g in OO.seeded_grads(model.named_parameters(), seed=s).items():
            dict(model.named_parameters())[n].grad = g
        opt.step()
        ema.update(model)
        after.append({"params": {n: OO.tensor_stats(p) for n, p in model.named_parameters()},
                      "ema": {k: OO.tensor_stats(v) for k, v in ema.ema.state_dict().items() if v.dtype.is_floating_point},
                      "ema_decay": ema.decay(ema.updates)})
    return {"phi": phi, "optimizer": optimizer_type, "lr": lr, "momentum": momentum, "weight_decay": wd,
            "groups": groups, "steps": after}


def schedules():
    out = []
    for kind, lr, mn, total in (("cos", 1e-2, 1e-4, 300), ("cos", 1e-3, 1e-5, 100), ("step", 1e-2, 1e-4, 300),
                                ("cos", 5e-2, 5e-4, 10)):
        f = get_lr_scheduler(kind, lr, mn, total)
        out.append({"kind": kind, "lr": lr, "min_lr": mn, "total": total, "values": [f(e) for e in range(total)]})
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for opt_type in ("sgd", "adam"):
        with open(os.path.join(OUT, f"optim_nano_{opt_type}.json"), "w") as f:
            json.dump(run("nano", opt_type), f)
    with open(os.path.join(OUT, "optim_lr_schedules.json"), "w") as f:
        json.dump(schedules(), f)
    print("wrote", [n for n in os.listdir(OUT) if n.startswith("optim_")])
