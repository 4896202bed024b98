"""Golden vectors for the input-format conversions, generated in the build container by IMPORTING the
reference: utils_seg/utils.py `preprocess_input` and `resize_image` (the letterbox used by prediction), utils/utils.py
`preprocess_input_radar`.  The label clamp + one-hot of YoloDataset.__getitem__ (utils/dataloader.py:96-105) sits in a
module that needs cv2 and cannot be imported here: those two arrays are produced by the same three numpy statements,
restated below, and are marked `restated` in the fixture.  Commits arrays only: tests/golden/formats_small.npz.
    python tools/make_golden_formats.py"""
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
from utils_seg.utils import preprocess_input, resize_image      # noqa: E402
from utils.utils import preprocess_input_radar                  # noqa: E402
from PIL import Image                                            # noqa: E402

rng = np.random.default_rng(20261003)
B, H, W, NS = 2, 12, 16, 9
img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
img[0, 0, 0] = (0, 128, 255)
png = rng.integers(0, NS + 1, (B, H, W), dtype=np.uint8)
png[1, 3, :5] = 255                                            # white borders of VOC-style labels -> ignore class
png[0, 0, 0] = NS
# the reference: float64 copy, preprocess_input in place, HWC -> CHW, FloatTensor (dataloader.py:88, 452)
images = np.stack([np.transpose(preprocess_input(np.array(im, dtype=np.float64)), [2, 0, 1]) for im in img]).astype(np.float32)
# restated (dataloader.py:96-105)
p2 = png.copy()
p2[p2 >= NS] = NS
onehot = np.eye(NS + 1)[p2.reshape([-1])].reshape((B, H, W, NS + 1)).astype(np.float32)
# radar min-max (prediction path)
radar = rng.standard_normal((4, H, W))
radar_n = preprocess_input_radar(radar.copy())
# letterbox geometry through the reference's resize_image on a 40 x 23 image into 32 x 32
src = Image.fromarray(rng.integers(0, 256, (23, 40, 3), dtype=np.uint8))
boxed, nw, nh = resize_image(src, (32, 32))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "formats_small.npz")
np.savez_compressed(out, img=img, png=png, num_classes_seg=NS, images=images, png_clamped=p2.astype(np.int64), onehot=onehot,
                    radar=radar, radar_norm=radar_n, letterbox_src=np.array(src), letterbox_out=np.array(boxed),
                    letterbox_nw_nh=np.array([nw, nh]))
print("wrote", out, os.path.getsize(out), "bytes")
