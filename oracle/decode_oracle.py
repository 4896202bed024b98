"""TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference's box decode.

`decode_outputs` follows utils/utils_bbox.py:32-84 line by line (that module cannot be imported here: it needs
torchvision and calls .cuda()); its xy / wh arithmetic is pinned by tests/golden/decode_small.npz, generated with the
importable `YOLOLoss.get_output_and_grid` (nets/yolo_training.py:99-111) by tools/make_golden_decode.py.
`yolo_correct_boxes` restates utils_bbox.py:5-30."""
import numpy as np
import torch


def decode_outputs(outputs, input_shape):
    hw = [x.shape[-2:] for x in outputs]
    out = torch.cat([x.flatten(start_dim=2) for x in outputs], dim=2).permute(0, 2, 1).clone()      # :44
    out[:, :, 4:] = torch.sigmoid(out[:, :, 4:])                                                     # :48
    grids, strides = [], []
    for h, w in hw:                                                                                  # :49-66
        gy, gx = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
        grid = torch.stack((gx, gy), 2).view(1, -1, 2)
        grids.append(grid)
        strides.append(torch.full((1, grid.shape[1], 1), input_shape[0] / h))
    grids = torch.cat(grids, dim=1).type(out.type())
    strides = torch.cat(strides, dim=1).type(out.type())
    out[..., :2] = (out[..., :2] + grids) * strides                                                  # :77
    out[..., 2:4] = torch.exp(out[..., 2:4]) * strides                                               # :78
    out[..., [0, 2]] = out[..., [0, 2]] / input_shape[1]                                             # :82
    out[..., [1, 3]] = out[..., [1, 3]] / input_shape[0]                                             # :83
    return out


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    box_yx, box_hw = box_xy[..., ::-1], box_wh[..., ::-1].copy()
    input_shape, image_shape = np.array(input_shape), np.array(image_shape)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2. / input_shape
        scale = input_shape / new_shape
        box_yx = (box_yx - offset) * scale
        box_hw *= scale
    mins, maxes = box_yx - box_hw / 2., box_yx + box_hw / 2.
    boxes = np.concatenate([mins[..., 0:1], mins[..., 1:2], maxes[..., 0:1], maxes[..., 1:2]], axis=-1)
    return boxes * np.concatenate([image_shape, image_shape], axis=-1)
