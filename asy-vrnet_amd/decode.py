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
    """Normalised (cx, cy), (w, h) of the network input -> (y1, x1, y2, x2) in pixels of the original image, undoing
    the letterbox when there was one (interface of utils/utils_bbox.py:5-30; host side, on the few boxes NMS keeps).
    input_shape / image_shape are (H, W)."""
    net_hw = np.asarray(input_shape, dtype=np.float64)
    img_hw = np.asarray(image_shape, dtype=np.float64)
    centre = np.asarray(box_xy, dtype=np.float64)[..., ::-1]          # (cy, cx)
    size = np.asarray(box_wh, dtype=np.float64)[..., ::-1]            # (h, w)
    if letterbox_image:
        # the image occupies a centred `inner` window of the network input: map window coordinates to [0, 1]
        inner = np.round(img_hw * (net_hw / img_hw).min())
        centre = (centre - 0.5 * (net_hw - inner) / net_hw) * (net_hw / inner)
        size = size * (net_hw / inner)
    top_left, bottom_right = centre - 0.5 * size, centre + 0.5 * size
    return np.concatenate([top_left * img_hw, bottom_right * img_hw], axis=-1)
