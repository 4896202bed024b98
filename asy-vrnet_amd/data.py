"""Input formats of the training pipeline, restated from the reference's dataloader (utils/dataloader.py is
not importable here: cv2 / albumentations).  Only the deterministic evaluation path (`random=False`) and the format
conversions are covered; the random augmentations (mosaic, mixup, HSV jitter, :217-437) are a training-recipe concern
outside the hot path.  Host side (numpy / PIL): parsing, letterbox, box mapping.  Device side (`device_batch`,
csrc/formats.hip): the per-pixel conversions of a letterboxed batch -- image normalisation + CHW, label clamp, one-hot --
from bytes.  Pins: the functions the reference keeps in importable modules (`preprocess_input`, `preprocess_input_radar`,
`resize_image`) produced tests/golden/formats_small.npz (tools/make_golden_formats.py); round 5: the reference's own
`YoloDataset.__getitem__` (train=False) + `yolo_dataset_collate`, importable once cv2 / albumentations are stubbed (the
evaluation path touches neither), produced tests/golden/dataset_small.npz (tools/make_golden_dataset.py) -- parsing,
letterbox, box mapping, clamp, one-hot and collate are held to it bit for bit (tests/test_data.py)."""
import os
import re

import numpy as np
import torch

MEAN = np.array([0.485, 0.456, 0.406])
STD = np.array([0.229, 0.224, 0.225])
_FRAME = re.compile(r"\d{10}.\d{5}")            # utils/dataloader.py:76-79: the frame id (timestamp) inside the path


def parse_annotation_line(line):
    """`path x1,y1,x2,y2,cls x1,y1,x2,y2,cls ...` (README.md:21-26; dataloader.py:119,133) -> (path, (n,5) int array)."""
    parts = line.split()
    if not parts:
        raise ValueError("empty annotation line")
    boxes = np.array([np.array(list(map(int, b.split(",")))) for b in parts[1:]])
    if boxes.size and boxes.shape[1] != 5:
        raise ValueError(f"annotation boxes need 5 comma-separated integers: {line!r}")
    return parts[0], boxes.reshape(-1, 5)


def frame_id(line):
    """Last `\\d{10}.\\d{5}` match of the line: names the radar .npz and the segmentation .png (dataloader.py:76-84)."""
    m = _FRAME.findall(line)
    if not m:
        raise ValueError(f"no frame id (\\d{{10}}.\\d{{5}}) in {line!r}")
    return m[-1]


def load_radar(radar_root, fid):
    """radar_root/<frame id>.npz, array `arr_0` of shape (4, H, W) (dataloader.py:111-112)."""
    return np.load(os.path.join(radar_root, fid + ".npz"))["arr_0"]


def preprocess_input(image):
    """utils_seg/utils.py:43-47 (HWC, RGB, 0..255 -> normalised); works on a float copy."""
    image = np.array(image, dtype=np.float64)
    image /= 255.0
    image -= MEAN
    image /= STD
    return image


def preprocess_input_radar(data):
    """utils/utils.py:50-53 (min-max to [0,1] + 1e-13)."""
    lo = np.min(data)
    return (data - lo) / (np.max(data) - lo) + 0.0000000000001


def letterbox_geometry(iw, ih, w, h):
    """dataloader.py:131-135: (nw, nh, dx, dy) of the aspect-preserving resize pasted into a w x h canvas."""
    scale = min(w / iw, h / ih)
    nw, nh = int(iw * scale), int(ih * scale)
    return nw, nh, (w - nw) // 2, (h - nh) // 2


def adjust_boxes(box, iw, ih, w, h):
    """Boxes (x1, y1, x2, y2, cls) of an iw x ih image -> the letterboxed w x h canvas: scaled and shifted like the
    pixels, clipped to the canvas, and boxes thinner than 2 px dropped (dataloader.py:170-180, minus its in-place
    shuffle).  The reference parses the annotation into an INTEGER array (:133) and assigns the scaled coordinates back
    into it, so every mapped coordinate is truncated towards zero before clipping -- reproduced here (pinned by
    tests/golden/dataset_small.npz, the reference's own output)."""
    nw, nh, dx, dy = letterbox_geometry(iw, ih, w, h)
    out = np.array(box, dtype=np.int64).reshape(-1, 5)
    if out.shape[0] == 0:
        return out.astype(np.float64)
    out[:, [0, 2]] = out[:, [0, 2]] * nw / iw + dx            # float result -> int64 storage: truncation, as the reference
    out[:, [1, 3]] = out[:, [1, 3]] * nh / ih + dy
    out[:, 0:2][out[:, 0:2] < 0] = 0
    out[:, 2][out[:, 2] > w] = w
    out[:, 3][out[:, 3] > h] = h
    keep = ((out[:, 2] - out[:, 0]) > 1) & ((out[:, 3] - out[:, 1]) > 1)
    return out[keep].astype(np.float64)


def boxes_xyxy_to_cxcywh(box):
    """dataloader.py:91-94: the [cx, cy, w, h, cls] rows YOLOLoss consumes."""
    box = np.array(box, dtype=np.float64).reshape(-1, 5)
    if len(box) != 0:
        box[:, 2:4] = box[:, 2:4] - box[:, 0:2]
        box[:, 0:2] = box[:, 0:2] + box[:, 2:4] / 2
    return box


def seg_targets(png, num_classes_seg):
    """dataloader.py:96-105: labels >= num_classes_seg become the ignore class; one-hot with the extra channel."""
    png = np.array(png)
    png[png >= num_classes_seg] = num_classes_seg
    seg_labels = np.eye(num_classes_seg + 1)[png.reshape([-1])].reshape(png.shape + (num_classes_seg + 1,))
    return png, seg_labels


def letterbox_sample(image, seg_label, box, input_shape):
    """The `random=False` branch of get_random_data (dataloader.py:137-183) on PIL images: bicubic resize onto a grey
    (128) canvas, nearest resize of the label onto a 0 canvas, boxes mapped alongside."""
    from PIL import Image
    iw, ih = image.size
    h, w = input_shape
    nw, nh, dx, dy = letterbox_geometry(iw, ih, w, h)
    new_image = Image.new("RGB", [w, h], (128, 128, 128))
    new_image.paste(image.convert("RGB").resize((nw, nh), Image.BICUBIC), (dx, dy))
    new_label = Image.new("L", [w, h], (0))
    new_label.paste(Image.fromarray(np.array(seg_label)).resize((nw, nh), Image.NEAREST), (dx, dy))
    return new_image, adjust_boxes(box, iw, ih, w, h), new_label


def resize_image(image, size):
    """utils_seg/utils.py:20-31 (the letterbox of the prediction scripts): bicubic resize onto a grey canvas.
    Returns (canvas, nw, nh)."""
    from PIL import Image
    iw, ih = image.size
    w, h = size
    nw, nh, dx, dy = letterbox_geometry(iw, ih, w, h)
    canvas = Image.new("RGB", size, (128, 128, 128))
    canvas.paste(image.resize((nw, nh), Image.BICUBIC), (dx, dy))
    return canvas, nw, nh


def device_batch(images_u8, pngs_u8, num_classes_seg, device="cuda"):
    """The tensors `yolo_dataset_collate` ships for a letterboxed batch, made ON THE DEVICE from bytes
    (vrnet_batch_formats_u8): images_u8 (B,H,W,3) uint8 RGB -> images (B,3,H,W) float32 normalised as preprocess_input does
    (bit-identical); pngs_u8 (B,H,W) uint8 -> png (B,H,W) int64 with the ignore class, seg_labels (B,H,W,nc+1) float32.
    numpy arrays or tensors; either input may be None.  4 B per pixel cross PCIe instead of 12 + 8 + 4 (nc + 1)."""
    from . import hip

    def dev(a):
        if a is None:
            return None
        t = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
        if t.dtype != torch.uint8:
            raise RuntimeError(f"device_batch: expected uint8 bytes, got {t.dtype}")
        return t.to(device, non_blocking=True).contiguous()
    return hip.batch_formats(dev(images_u8), dev(pngs_u8), num_classes_seg)


def make_sample(image, box, radar, png, num_classes_seg):
    """YoloDataset.__getitem__ after augmentation (dataloader.py:88-107): (image CHW float64, boxes cxcywh, radar,
    png, one-hot)."""
    image = np.transpose(preprocess_input(np.array(image, dtype=np.float64)), [2, 0, 1])
    png, seg_labels = seg_targets(png, num_classes_seg)
    return image, boxes_xyxy_to_cxcywh(box), np.array(radar, dtype=np.float64), png, seg_labels


def yolo_dataset_collate(batch):
    """dataloader.py:440-457."""
    images, bboxes, radars, pngs, seg_labels = zip(*batch)
    return (torch.from_numpy(np.array(images)).type(torch.FloatTensor),
            [torch.from_numpy(np.array(a, dtype=np.float64).reshape(-1, 5)).type(torch.FloatTensor) for a in bboxes],
            torch.from_numpy(np.array(radars)).type(torch.FloatTensor),
            torch.from_numpy(np.array(pngs)).long(),
            torch.from_numpy(np.array(seg_labels)).type(torch.FloatTensor))
