"""Box decode of the det maps, mirroring utils/utils_bbox.py: `decode_outputs` runs as one HIP kernel
(no concat / grid / stride tensors); `yolo_correct_boxes` is the reference's host-side letterbox un-map (numpy, on
the handful of boxes that survive NMS).  NMS itself is torchvision's `batched_nms` in the reference
(utils_bbox.py:124) and is out of scope (parity unpinned)."""
import numpy as np
import torch

from . import hip


def decode_outputs(outputs, input_shape, local_rank=None):
    """outputs: list of (B, 5+nc, h, w) raw head maps on the GPU (P3, P4, P5); input_shape = (H, W).
    Returns (B, sum h*w, 5+nc): normalised cx, cy, w, h, then sigmoid(obj), sigmoid(cls...).  utils_bbox.py:32-84
    (`local_rank` is accepted for signature parity; the result lives on the inputs' device)."""
    outs = [o.detach().contiguous().float() for o in outputs]
    if not outs or not all(o.is_cuda and o.dim() == 4 and o.shape[:2] == outs[0].shape[:2] for o in outs):
        raise RuntimeError("decode_outputs: expects a list of (B, 5+nc, h, w) GPU tensors of one batch")
    B, C = outs[0].shape[:2]
    A = sum(o.shape[2] * o.shape[3] for o in outs)
    out = torch.empty((B, A, C), dtype=torch.float32, device=outs[0].device)
    hip.decode_outputs(outs, input_shape[0], input_shape[1], out)
    return out


def yolo_correct_boxes(box_xy, box_wh, input_shape, image_shape, letterbox_image):
    """utils_bbox.py:5-30: normalised centre/size -> (y1, x1, y2, x2) in pixels of the original image."""
    box_yx = box_xy[..., ::-1]
    box_hw = box_wh[..., ::-1]
    input_shape = np.array(input_shape)
    image_shape = np.array(image_shape)
    if letterbox_image:
        new_shape = np.round(image_shape * np.min(input_shape / image_shape))
        offset = (input_shape - new_shape) / 2. / input_shape
        scale = input_shape / new_shape
        box_yx = (box_yx - offset) * scale
        box_hw = box_hw * scale
    box_mins = box_yx - (box_hw / 2.)
    box_maxes = box_yx + (box_hw / 2.)
    boxes = np.concatenate([box_mins[..., 0:1], box_mins[..., 1:2], box_maxes[..., 0:1], box_maxes[..., 1:2]], axis=-1)
    boxes = boxes * np.concatenate([image_shape, image_shape], axis=-1)
    return boxes
