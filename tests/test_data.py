"""Input formats: every function against hand-worked values, against the functions the reference keeps in
importable modules (formats_small.npz) and -- round 5 -- against the reference's own YoloDataset + collate (dataset_small.npz)."""
import os

import numpy as np
import pytest
import torch

from asy_vrnet_amd import data

LINE = "/data/VOC2007/JPEGImages/1664091257.87023.jpg 10,20,110,220,3 0,0,5,5,1"


def test_annotation_line_and_frame_id():
    path, boxes = data.parse_annotation_line(LINE)
    assert path.endswith("1664091257.87023.jpg") and boxes.tolist() == [[10, 20, 110, 220, 3], [0, 0, 5, 5, 1]]
    assert data.frame_id(LINE) == "1664091257.87023"
    p, b = data.parse_annotation_line("/x/1664091257.87023.jpg")
    assert b.shape == (0, 5)
    with pytest.raises(ValueError):
        data.parse_annotation_line("/x/a.jpg 1,2,3,4")
    with pytest.raises(ValueError):
        data.frame_id("/x/a.jpg")


def test_radar_npz_roundtrip(tmp_path):
    arr = np.arange(4 * 6 * 8, dtype=np.float32).reshape(4, 6, 8)
    np.savez(os.path.join(tmp_path, "1664091257.87023.npz"), arr)
    got = data.load_radar(str(tmp_path), data.frame_id(LINE))
    assert got.shape == (4, 6, 8) and np.array_equal(got, arr)
    r = data.preprocess_input_radar(arr)
    assert abs(r.min() - 1e-13) < 1e-15 and abs(r.max() - 1.0) < 1e-9


def test_image_normalisation():
    img = np.zeros((2, 2, 3))
    img[0, 0] = [255, 255, 255]
    out = data.preprocess_input(img)
    assert np.allclose(out[0, 0], (1 - data.MEAN) / data.STD) and np.allclose(out[1, 1], -data.MEAN / data.STD)
    assert img[0, 0, 0] == 255                                    # the input is not modified


def test_letterbox_boxes_and_cxcywh():
    # 640x360 image into 512x512: scale .8 -> 512x288, dy = 112
    assert data.letterbox_geometry(640, 360, 512, 512) == (512, 288, 0, 112)
    box = np.array([[100, 50, 300, 250, 2], [630, 350, 640, 360, 1], [0, 0, 1, 1, 0]])
    out = data.adjust_boxes(box, 640, 360, 512, 512)
    assert np.allclose(out, [[80, 152, 240, 312, 2], [504, 392, 512, 400, 1]])     # 0.8 px wide box dropped
    cx = data.boxes_xyxy_to_cxcywh(out)
    assert np.allclose(cx, [[160, 232, 160, 160, 2], [508, 396, 8, 8, 1]])
    assert data.boxes_xyxy_to_cxcywh(np.zeros((0, 5))).shape == (0, 5)


def test_seg_targets_and_collate():
    png = np.array([[0, 3], [9, 255]])
    p, oh = data.seg_targets(png, 9)
    assert p.tolist() == [[0, 3], [9, 9]] and oh.shape == (2, 2, 10)
    assert oh[1, 1].tolist() == [0] * 9 + [1] and oh[0, 1, 3] == 1
    s1 = data.make_sample(np.full((2, 2, 3), 255.0), np.array([[0, 0, 2, 2, 1]]), np.ones((4, 2, 2)), png, 9)
    s2 = data.make_sample(np.zeros((2, 2, 3)), np.zeros((0, 5)), np.zeros((4, 2, 2)), png, 9)
    images, boxes, radars, pngs, seg = data.yolo_dataset_collate([s1, s2])
    assert images.shape == (2, 3, 2, 2) and images.dtype == torch.float32
    assert boxes[0].tolist() == [[1, 1, 2, 2, 1]] and boxes[1].shape == (0, 5)
    assert radars.shape == (2, 4, 2, 2) and pngs.dtype == torch.int64 and seg.shape == (2, 2, 2, 10)


def test_letterbox_sample_with_pil():
    Image = pytest.importorskip("PIL.Image")
    img = Image.fromarray(np.full((36, 64, 3), 200, dtype=np.uint8))
    lab = Image.fromarray(np.full((36, 64), 5, dtype=np.uint8))
    new_image, box, new_label = data.letterbox_sample(img, lab, np.array([[10, 5, 30, 25, 2]]), (64, 64))
    a, l = np.array(new_image), np.array(new_label)
    assert a.shape == (64, 64, 3) and (a[0, 0] == 128).all() and (a[32, 32] == 200).all()      # grey bars, image centre
    assert l[0, 0] == 0 and l[32, 32] == 5 and np.allclose(box, [[10, 19, 30, 39, 2]])


def test_reference_annotation_file_sample():
    """First lines of the reference's own 2007_val.txt (a data file it ships), copied verbatim as a fixture."""
    here = os.path.dirname(os.path.abspath(__file__))
    lines = open(os.path.join(here, "golden", "annotation_lines_sample.txt")).read().splitlines()
    assert len(lines) == 5
    path, boxes = data.parse_annotation_line(lines[0])
    assert path.endswith("1664091274.85923.jpg") and data.frame_id(lines[0]) == "1664091274.85923"
    assert boxes.tolist() == [[889, 405, 903, 450, 0], [1001, 404, 1015, 458, 0], [1, 396, 23, 438, 0]]
    for ln in lines:
        p, b = data.parse_annotation_line(ln)
        assert b.shape[1] == 5 and (b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all() and data.frame_id(ln) in p
        cx = data.boxes_xyxy_to_cxcywh(data.adjust_boxes(b, 1920, 1080, 512, 512))
        assert (cx[:, 2:4] > 1).all() and (cx[:, :2] >= 0).all() and (cx[:, :2] <= 512).all()


# ---- pinned by the reference's own importable functions (tools/make_golden_formats.py -> formats_small.npz) ----------
def _golden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "formats_small.npz"))


def test_host_formats_match_reference_vectors():
    z = _golden()
    ns = int(z["num_classes_seg"])
    imgs = np.stack([np.transpose(data.preprocess_input(im), [2, 0, 1]) for im in z["img"]]).astype(np.float32)
    assert np.array_equal(imgs, z["images"])                       # preprocess_input, utils_seg/utils.py:43-47
    for b in range(z["png"].shape[0]):
        png, onehot = data.seg_targets(z["png"][b], ns)
        assert np.array_equal(png, z["png_clamped"][b]) and np.array_equal(onehot.astype(np.float32), z["onehot"][b])
    assert np.array_equal(data.preprocess_input_radar(z["radar"]), z["radar_norm"])      # utils/utils.py:50-53
    from PIL import Image
    boxed, nw, nh = data.resize_image(Image.fromarray(z["letterbox_src"]), (32, 32))     # utils_seg/utils.py:20-31
    assert [nw, nh] == z["letterbox_nw_nh"].tolist() and np.array_equal(np.array(boxed), z["letterbox_out"])


@pytest.mark.gpu
def test_device_batch_formats_bit_identical():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    z = _golden()
    ns = int(z["num_classes_seg"])
    images, png, onehot = data.device_batch(z["img"], z["png"], ns)
    assert images.dtype == torch.float32 and png.dtype == torch.int64 and onehot.dtype == torch.float32
    assert torch.equal(images.cpu(), torch.from_numpy(z["images"]))
    assert torch.equal(png.cpu(), torch.from_numpy(z["png_clamped"]))
    assert torch.equal(onehot.cpu(), torch.from_numpy(z["onehot"]))
    # either half alone; what the collate function of the host path produces for the same samples
    im2, p2, o2 = data.device_batch(z["img"], None, ns)
    assert p2 is None and o2 is None and torch.equal(im2, images)
    _, p3, o3 = data.device_batch(None, z["png"], ns)
    assert torch.equal(p3, png) and torch.equal(o3, onehot)
    batch = [data.make_sample(z["img"][b], np.zeros((0, 5)), z["radar"], z["png"][b], ns) for b in range(2)]
    h_images, _, _, h_png, h_onehot = data.yolo_dataset_collate(batch)
    assert torch.equal(h_images, images.cpu()) and torch.equal(h_png, png.cpu()) and torch.equal(h_onehot, onehot.cpu())
    # a full-size batch: every pixel value class through the kernel, against the 256-entry table of the host function
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, (8, 512, 512, 3), dtype=np.uint8)
    lab = rng.integers(0, 256, (8, 512, 512), dtype=np.uint8)
    bi, bp, bo = data.device_batch(big, lab, ns)
    lut = np.stack([data.preprocess_input(np.full((1, 1, 3), v))[0, 0] for v in range(256)]).astype(np.float32)      # (256, 3)
    want = np.stack([lut[big[..., c], c] for c in range(3)], 1)
    assert np.array_equal(bi.cpu().numpy(), want)
    want_p = np.minimum(lab, ns).astype(np.int64)
    assert np.array_equal(bp.cpu().numpy(), want_p)
    assert np.array_equal(bo.cpu().numpy().argmax(-1), want_p) and float(bo.sum()) == lab.size
    with pytest.raises(RuntimeError, match="uint8"):
        data.device_batch(big.astype(np.float32), None, ns)


def test_dataset_item_and_collate_match_the_reference(tmp_path):
    """Row f4 end to end against the reference's OWN YoloDataset.__getitem__ (train=False) + yolo_dataset_collate
    (utils/dataloader.py:71-183, 440-457), run in the build container with cv2 / albumentations stubbed
    (tools/make_golden_dataset.py -> tests/golden/dataset_small.npz): annotation parsing, frame id, radar .npz, bicubic
    letterbox, nearest label letterbox, box mapping incl. the reference's integer truncation, clipping and thin-box filter,
    cx-cy-w-h, label clamp to the ignore class, one-hot, collate dtypes -- bit for bit (the boxes up to row order: the
    reference shuffles them in place)."""
    from PIL import Image
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_small.npz"))
    h, w = (int(v) for v in z["input_shape"])
    ns = int(z["num_classes_seg"])
    os.makedirs(tmp_path / "radar")
    batch = []
    for i, tail in enumerate(z["line_tails"]):
        fid = str(z[f"fid{i}"])
        img_path = str(tmp_path / (fid + ".png"))
        Image.fromarray(z[f"img{i}"]).save(img_path)
        np.savez(tmp_path / "radar" / (fid + ".npz"), z[f"radar{i}"])
        line = img_path + (" " + str(tail) if str(tail) else "")
        path, boxes = data.parse_annotation_line(line)
        assert path == img_path and data.frame_id(line) == fid
        radar = data.load_radar(str(tmp_path / "radar"), data.frame_id(line))
        image, box, label = data.letterbox_sample(Image.open(path), Image.fromarray(z[f"lab{i}"]), boxes, (h, w))
        batch.append(data.make_sample(image, box, radar, label, ns))
    images, bboxes, radars, pngs, seg_labels = data.yolo_dataset_collate(batch)
    assert images.dtype == torch.float32 and pngs.dtype == torch.int64 and seg_labels.dtype == torch.float32
    assert np.array_equal(images.numpy(), z["images"])
    assert np.array_equal(radars.numpy(), z["radars"])
    assert np.array_equal(pngs.numpy(), z["pngs"])
    assert np.array_equal(seg_labels.numpy(), z["seg_labels"])
    for i, b in enumerate(bboxes):
        want = z[f"boxes_out{i}"].reshape(-1, 5)      # (the reference returns shape (0,) for a frame without boxes)
        got = b.numpy()
        assert got.shape == want.shape, (i, got, want)
        order = lambda a: a[np.lexsort(a.T[::-1])]
        assert np.array_equal(order(got), order(want)), (i, got, want)
