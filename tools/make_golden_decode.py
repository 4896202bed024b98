"""Golden vectors for the box decode.  utils/utils_bbox.py is not importable in the build container
(torchvision, hard .cuda()), so the fixture pins what IS importable from the reference:
`YOLOLoss.get_output_and_grid` (nets/yolo_training.py:99-111) gives the un-normalised (xy + grid) * stride and
exp(wh) * stride of every level -- the same arithmetic decode_outputs applies (utils_bbox.py:77-78) -- and
torch.sigmoid gives channels >= 4; the two normalising divisions (:82-83) are applied here as written there.

    python tools/make_golden_decode.py       # writes tests/golden/decode_small.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_harness import load_reference  # noqa: E402

load_reference()
from nets.yolo_training import YOLOLoss  # noqa: E402

if __name__ == "__main__":
    rng = np.random.default_rng(2024)
    B, nc, H, W = 2, 4, 64, 96                     # non-square input: exercises stride = input_h / h on both axes
    shapes = [(H // 8, W // 8), (H // 16, W // 16), (H // 32, W // 32)]
    levels = [torch.from_numpy(rng.standard_normal((B, 5 + nc, h, w)).astype(np.float32)) for h, w in shapes]
    yl = YOLOLoss(nc, fp16=False, strides=[8, 16, 32])
    outs = []
    for k, (stride, lv) in enumerate(zip(yl.strides, levels)):
        o, _ = yl.get_output_and_grid(lv.clone(), k, stride)          # (B, hw, C) with xy, wh decoded in pixels
        outs.append(o)
    out = torch.cat(outs, 1)
    out[..., 4:] = torch.sigmoid(out[..., 4:])
    out[..., [0, 2]] = out[..., [0, 2]] / W
    out[..., [1, 3]] = out[..., [1, 3]] / H
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "decode_small.npz"), input_shape=np.array([H, W]),
                        decoded=out.numpy(), **{f"level{i}": lv.numpy() for i, lv in enumerate(levels)})
    print("wrote decode_small.npz", out.shape)
