"""The drop-in boundary's state_dict surface: 887 keys with the reference's
names, shapes, dtypes and order; fixtures written by tools/make_golden.py from the reference."""
import json
import os

import pytest
import torch

import asy_vrnet_amd as A


@pytest.mark.parametrize("phi", ["nano", "l"])
def test_state_dict_matches_reference_surface(phi, golden_dir):
    surf = json.load(open(os.path.join(golden_dir, f"state_dict_surface_{phi}.json")))
    m = A.EfficientVRNet(num_classes=4, num_seg_classes=9, phi=phi)
    sd = m.state_dict()
    assert len(sd) == len(surf) == 887
    mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
    assert mine == surf


def test_fea_pos_values():
    m = A.EfficientVRNet(4, 9, "nano", img_size=64)
    pos = m.backbone.backbone.fea_pos
    assert pos.shape == (64, 64, 2)
    assert torch.allclose(pos[5, 9], torch.tensor([5 / 63.0 - 0.5, 9 / 63.0 - 0.5]))
    assert torch.equal(pos, m.backbone.backbone.fea_pos_r)


def test_caller_introspection_contracts():
    """train.py:460-473 optimizer grouping and yolo_training.py:482-501 weights_init."""
    m = A.EfficientVRNet(4, 9, "nano")
    pg0, pg1, pg2 = [], [], []
    for k, v in m.named_modules():
        if hasattr(v, "bias") and isinstance(v.bias, torch.nn.Parameter):
            pg2.append(v.bias)
        if isinstance(v, torch.nn.BatchNorm2d) or "bn" in k:
            pg0.append(v.weight)
        elif hasattr(v, "weight") and isinstance(v.weight, torch.nn.Parameter):
            pg1.append(v.weight)
    n_opt = len({id(p) for p in pg0 + pg1 + pg2})
    n_all = len(list(m.parameters()))
    assert n_all == 735                         # incl. 6 zero-sized ShuffleAttention(channel=3) entries
    assert n_opt == 595                         # measured on the reference: 140 never reach the optimizer
    for mod in m.modules():                     # containers must not look like Conv leaves
        if "Conv" in type(mod).__name__ and not isinstance(mod, (torch.nn.Conv2d, torch.nn.Conv1d)):
            assert not hasattr(mod, "weight")
    assert len(list(m.backbone.backbone.parameters())) > 0
