"""Third-party anchors for two of the stages that cannot be pinned to the reference (their code lives in OpenCV, which is
not installed anywhere we run): fixtures produced by scikit-image 0.18.3 in this container's conda interpreter
(tests/golden/make_thirdparty.py).  scikit-image's FAST is its own implementation, not OpenCV's and not ours.  CPU only."""
import os

import numpy as np

from orb_slam2_ssd_semantic_amd.synth import synth_frame

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "thirdparty_skimage.npz"))


def test_fast9_corner_classification_equals_skimage(oracle):
    """cv::FAST's segment test: a pixel is a corner at threshold t  <=>  oracle score >= t, on the
    detectable interior; 8 images x thresholds, ~30 000 corners, every pixel must agree with skimage.corner_fast(n=9)."""
    total = 0
    for seed in range(4):
        img = G[f"fast/img{seed}"]
        assert np.array_equal(img, synth_frame(seed, 96, 128))      # the fixture's images are the repo's generator
        sc = oracle.fast_score_map(img)
        for t in (7, 20):
            theirs = np.unpackbits(G[f"fast/corners{seed}_t{t}"])[:img.size].reshape(img.shape).astype(bool)
            ours = np.zeros_like(theirs)
            ours[3:-3, 3:-3] = sc[3:-3, 3:-3] >= t
            assert np.array_equal(ours, theirs), (seed, t)
            total += int(theirs.sum())
            # and through the oracle's cv::FAST itself (no NMS): the same set of positions
            k = oracle.fast9(img, t, nonmax=False)
            got = np.zeros_like(theirs)
            got[k["y"].astype(int), k["x"].astype(int)] = True
            assert np.array_equal(got, theirs), (seed, t)
    assert total > 25000


def test_brief_pattern_and_orientation_mask_equal_skimage(oracle):
    """the 256 x 4 rBRIEF table (src/ORBextractor.cc:135-393) and umax / the 749-pixel circular patch (:449-465, :59-88)
    against the copies scikit-image ships (taken by its authors from OpenCV's orb.cpp)"""
    assert np.array_equal(G["orb/positions"].astype(np.int32), oracle.pattern().reshape(256, 4).astype(np.int32))
    assert np.array_equal(G["orb/umax"], oracle.umax())
    mask = G["orb/ofast_mask"]
    assert mask.sum() == 749
    um = oracle.umax()
    ours = np.zeros((31, 31), np.uint8)
    for v in range(-15, 16):
        ours[15 + v, 15 - um[abs(v)]:15 + um[abs(v)] + 1] = 1
    assert np.array_equal(ours, mask) and np.array_equal(ours, ours.T)
    # the moments the oracle computes are the masked sums of that patch
    img = synth_frame(5, 96, 128).astype(np.int64)
    uu, vv = np.meshgrid(np.arange(-15, 16), np.arange(-15, 16))
    for (x, y) in ((40, 30), (100, 70), (19, 19)):
        p = img[y - 15:y + 16, x - 15:x + 16] * mask
        m10, m01 = oracle.ic_moments(img.astype(np.uint8), x, y)
        assert (m10, m01) == (int((p * uu).sum()), int((p * vv).sum()))
