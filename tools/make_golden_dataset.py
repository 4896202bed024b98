"""Golden vectors for the rest of row f4: annotation parsing, letterbox + box mapping, label clamp, one-hot and
collate -- produced by the reference's OWN `YoloDataset.__getitem__` (evaluation path, train=False) and
`yolo_dataset_collate` (utils/dataloader.py:71-107, 108-183, 440-457), imported in the build container.  The module imports
cv2 and albumentations at the top (neither is installed here) but the code path above touches neither: two throw-away stub
packages in a temporary directory satisfy the imports (cv2: empty; albumentations: Compose / Random* that are never called).
A tiny synthetic dataset (three frames of different aspect ratios, boxes that leave the canvas, a frame without boxes,
labels with VOC-style white borders) is written to a temporary directory in the formats the reference reads; the fixture
holds the raw inputs (arrays + annotation lines) and the reference's outputs.  Arrays only: tests/golden/dataset_small.npz.
    python tools/make_golden_dataset.py"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"


def write_stubs(d):
    os.makedirs(os.path.join(d, "cv2"))
    open(os.path.join(d, "cv2", "__init__.py"), "w").write("")
    os.makedirs(os.path.join(d, "albumentations"))
    open(os.path.join(d, "albumentations", "__init__.py"), "w").write(
        "class _T:\n    def __init__(self, *a, **k):\n        pass\n"
        "Compose = RandomRain = RandomSunFlare = RandomFog = _T\n")


def synthetic_frames(rng):
    """(frame id, image HWC uint8, label HW uint8, radar (4,H,W) float64, boxes (n,5) int) x 3."""
    frames = []
    for i, (ih, iw) in enumerate(((30, 52), (47, 33), (40, 40))):
        fid = f"16640{i:05d}.{12345 + i:05d}"
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        lab = rng.integers(0, 9, (ih, iw), dtype=np.uint8)
        lab[0, : iw // 2] = 255                                  # white border -> ignore class
        lab[1, 0] = 9
        radar = rng.standard_normal((4, 32, 32))
        if i == 0:
            boxes = np.array([[3, 4, 30, 20, 1], [-5, 2, 12, 29, 0], [40, 10, 60, 28, 3], [7, 7, 8, 25, 2]])   # one off-canvas, one thin
        elif i == 1:
            boxes = np.zeros((0, 5), dtype=np.int64)              # a frame without objects
        else:
            boxes = np.array([[0, 0, 40, 40, 2], [11, 13, 27, 35, 1]])
        frames.append((fid, img, lab, radar, boxes))
    return frames


def main():
    from PIL import Image
    rng = np.random.default_rng(20261004)
    frames = synthetic_frames(rng)
    with tempfile.TemporaryDirectory() as tmp:
        write_stubs(os.path.join(tmp, "stubs"))
        sys.path[:0] = [os.path.join(tmp, "stubs"), REF]
        sys.dont_write_bytecode = True
        from utils.dataloader import YoloDataset, yolo_dataset_collate
        root = os.path.join(tmp, "data")
        os.makedirs(os.path.join(root, "VOC2007", "SegmentationClass"))
        os.makedirs(os.path.join(root, "VOC2007", "JPEGImages"))
        os.makedirs(os.path.join(root, "radar"))
        lines = []
        for fid, img, lab, radar, boxes in frames:
            path = os.path.join(root, "VOC2007", "JPEGImages", fid + ".png")      # lossless: the fixture holds the decoded pixels
            Image.fromarray(img).save(path)
            Image.fromarray(lab).save(os.path.join(root, "VOC2007", "SegmentationClass", fid + ".png"))
            np.savez(os.path.join(root, "radar", fid + ".npz"), radar)
            lines.append(path + "".join(" " + ",".join(str(int(v)) for v in b) for b in boxes))
        input_shape = (32, 32)
        ds = YoloDataset(lines, input_shape, 4, 9, 1, os.path.join(root, "radar"), False, False, 0.0, 0.0, root, False)
        np.random.seed(7)                                         # get_random_data shuffles the boxes in place
        batch = [ds[i] for i in range(len(lines))]
        images, bboxes, radars, pngs, seg_labels = yolo_dataset_collate(batch)
    out = {"input_shape": np.array(input_shape), "num_classes_seg": 9, "images": images.numpy(), "radars": radars.numpy(),
           "pngs": pngs.numpy(), "seg_labels": seg_labels.numpy(),
           # the reference opens files by path: only the part of each line behind the path is data
           "line_tails": np.array([ln.split(" ", 1)[1] if " " in ln else "" for ln in lines])}
    for i, (fid, img, lab, radar, boxes) in enumerate(frames):
        out.update({f"fid{i}": fid, f"img{i}": img, f"lab{i}": lab, f"radar{i}": radar, f"boxes_out{i}": bboxes[i].numpy()})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "dataset_small.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes; boxes:", [tuple(b.shape) for b in bboxes])
    for b in bboxes:
        print(b)


if __name__ == "__main__":
    main()
