"""Box decode: oracle pinned against the reference's importable decode arithmetic; HIP kernel against
the oracle and the same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle import decode_oracle as DO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_small.npz")


def load():
    z = np.load(GOLD)
    levels = [torch.from_numpy(z[f"level{i}"]) for i in range(3)]
    return levels, tuple(int(v) for v in z["input_shape"]), torch.from_numpy(z["decoded"])


def test_oracle_matches_reference_fixture():
    levels, shape, want = load()
    got = DO.decode_outputs(levels, shape)
    assert got.shape == want.shape == (2, 8 * 12 + 4 * 6 + 2 * 3, 9)
    assert torch.allclose(got, want, rtol=1e-6, atol=1e-7)
    # stride is input_h / h on BOTH axes (utils_bbox.py:65): with H=64, W=96 the last anchor of level 0 sits at x = 11
    assert abs(got[0, 8 * 12 - 1, 0].item() - (levels[0][0, 0, 7, 11].item() + 11) * 8 / 96) < 1e-6


def test_yolo_correct_boxes_host_logic():
    from asy_vrnet_amd import decode
    xy = np.array([[0.5, 0.5], [0.25, 0.75]])
    wh = np.array([[0.2, 0.4], [0.1, 0.1]])
    for letterbox in (False, True):
        got = decode.yolo_correct_boxes(xy.copy(), wh.copy(), (512, 512), (360, 640), letterbox)
        assert np.allclose(got, DO.yolo_correct_boxes(xy.copy(), wh.copy(), (512, 512), (360, 640), letterbox))
    # by hand: letterbox of a 360x640 image into 512x512 scales by 0.8 -> 288x512, 112 px of padding above and below;
    # a centred box of normalised size (w .2, h .4) is 128 x 256 px wide in the original 640-wide image
    b = decode.yolo_correct_boxes(xy[:1].copy(), wh[:1].copy(), (512, 512), (360, 640), True)[0]
    assert np.allclose(b, [180 - 128, 320 - 64, 180 + 128, 320 + 64])
    b = decode.yolo_correct_boxes(xy[:1].copy(), wh[:1].copy(), (512, 512), (360, 640), False)[0]
    assert np.allclose(b, [180 - 72, 320 - 64, 180 + 72, 320 + 64])


@pytest.mark.gpu
def test_decode_kernel_matches_oracle_and_fixture():
    from asy_vrnet_amd import decode
    levels, shape, want = load()
    got = decode.decode_outputs([lv.cuda() for lv in levels], shape).cpu()
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-7)
    assert torch.allclose(got, DO.decode_outputs(levels, shape), rtol=2e-6, atol=1e-7)
    # the hot path's own det maps: level order P3, P4, P5 (coc_fpn_dual.py:224), bs 2 at 128 px
    rng = np.random.default_rng(5)
    big = [torch.from_numpy(rng.standard_normal((3, 9, s, s)).astype(np.float32)) for s in (64, 32, 16)]
    got = decode.decode_outputs([b.cuda() for b in big], (512, 512)).cpu()
    assert torch.allclose(got, DO.decode_outputs(big, (512, 512)), rtol=2e-6, atol=1e-7)
    with pytest.raises(RuntimeError):
        decode.decode_outputs([big[0].cuda(), big[1][:2].cuda()], (512, 512))
