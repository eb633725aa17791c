"""Input side of BASELINE configs 1-3: association list, cv::imread memory order, Tracking's colour -> gray conversion
(src/Tracking.cc:339-353).  The dataset itself is not in the container: a miniature sequence with the
TUM layout is written to a temp dir."""
import os

import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd import tum


def test_gray_formula_known_values():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [12, 200, 77]]], np.uint8)
    # memory order (c0, c1, c2) = cv::imread's (B, G, R); Camera.RGB = 1 -> CV_RGB2GRAY treats c0 as R
    assert tum.gray_from_interleaved(px, True).tolist() == [[76, 150, 29, 255, 0, (4899 * 12 + 9617 * 200 + 1868 * 77 + 8192) >> 14]]
    assert tum.gray_from_interleaved(px, False).tolist() == [[29, 150, 76, 255, 0, (1868 * 12 + 9617 * 200 + 4899 * 77 + 8192) >> 14]]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    ref = np.array([[(4899 * int(p[0]) + 9617 * int(p[1]) + 1868 * int(p[2]) + 8192) >> 14 for p in row] for row in img], np.uint8)
    assert np.array_equal(tum.gray_from_interleaved(img, True), ref)


def test_sequence_loader(tmp_path, monkeypatch):
    from PIL import Image
    rng = np.random.default_rng(1)
    os.makedirs(tmp_path / "rgb")
    names, imgs = [], []
    for i in range(4):
        rgb = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
        name = f"rgb/{1341846313.5 + i * 0.03:.6f}.png"
        Image.fromarray(rgb, "RGB").save(tmp_path / name)
        names.append(name)
        imgs.append(rgb)
    with open(tmp_path / "associate.txt", "w") as f:
        f.write("# comment\n\n")
        for i in (2, 0, 3):   # association order, not directory order
            f.write(f"{1341846313.5 + i * 0.03:.6f} {names[i]} {1341846313.6 + i:.6f} depth/x.png\n")
    monkeypatch.setenv("TUM_FR3_WALKING_XYZ", str(tmp_path))
    assert tum.sequence_dir() == str(tmp_path)
    rows = tum.load_associations(tmp_path / "associate.txt")
    assert [r[1] for r in rows] == [names[2], names[0], names[3]]
    g = tum.load_gray_frames()
    assert g.shape == (3, 48, 64)
    for out, i in zip(g, (2, 0, 3)):
        bgr = imgs[i][..., ::-1]                       # cv::imread memory order
        assert np.array_equal(tum.read_bgr(tmp_path / names[i]), bgr)
        assert np.array_equal(out, tum.gray_from_interleaved(bgr, True))
    assert tum.load_gray_frames(limit=2).shape[0] == 2
    os.remove(tmp_path / "associate.txt")
    with open(tmp_path / "rgb.txt", "w") as f:
        f.write("# color images\n" + "".join(f"{1341846313.5 + i * 0.03:.6f} {names[i]}\n" for i in range(4)))
    assert tum.load_gray_frames().shape[0] == 4       # no association file: the dataset's own rgb.txt
    monkeypatch.delenv("TUM_FR3_WALKING_XYZ")
    assert tum.sequence_dir() is None


@pytest.mark.gpu
def test_device_gray_conversion_equals_host():
    import torch
    from orb_slam2_ssd_semantic_amd import _ffi
    rng = np.random.default_rng(2)
    for (B, h, w, pad) in ((3, 480, 640, 0), (2, 37, 53, 5), (1, 8, 1030, 2)):
        stride = 3 * w + pad
        src = rng.integers(0, 256, (B, h, stride), dtype=np.uint8)
        d_src = torch.from_numpy(src).cuda()
        gstride = w + 3
        d_gray = torch.zeros((B, h, gstride), dtype=torch.uint8, device="cuda")
        for flag in (1, 0):
            rc = _ffi.lib().orbfe_interleaved_to_gray_device(d_src.data_ptr(), B, w, h, stride, h * stride, flag, d_gray.data_ptr(),
                                                             gstride, h * gstride, None)
            assert rc == 0
            torch.cuda.synchronize()
            got = d_gray.cpu().numpy()[:, :, :w]
            for b in range(B):
                img3 = src[b, :, :3 * w].reshape(h, w, 3)
                assert np.array_equal(got[b], tum.gray_from_interleaved(img3, bool(flag))), (B, h, w, flag)
