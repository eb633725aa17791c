"""Third-party anchors for the stages that cannot be pinned to the reference (their code lives in OpenCV, which is not
installed anywhere we run): fixtures produced by scikit-image 0.18.3 in this container's conda interpreter
(tests/golden/make_thirdparty.py) -- scikit-image's FAST is its own implementation, not OpenCV's and not ours -- and, below,
numpy / scipy run live for the border, the blur, the resize and fastAtan2.  CPU only."""
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


# ---- the other restated OpenCV primitives against numpy / scipy (independent code, present in this interpreter) -------------
# These are not bit-for-bit implementations of OpenCV's fixed-point paths, so the statements are the ones the published
# definitions allow: exact where the definition is exact (border), within one grey level of the real-valued definition
# and equal on the great majority of pixels where OpenCV rounds a fixed-point approximation (blur, resize), within the
# documented 0.3 degrees for fastAtan2.


def test_reflect101_border_equals_numpy_pad(oracle):
    """cv::copyMakeBorder(..., BORDER_REFLECT_101) (src/ORBextractor.cc:1133-1141) == numpy.pad(mode="reflect")."""
    for seed, (h, w) in enumerate(((40, 56), (21, 33), (20, 20))):
        img = synth_frame(seed, h, w)
        assert np.array_equal(oracle.copy_make_border101(img, 19), np.pad(img, 19, mode="reflect"))


def test_gaussian_blur_equals_scipy_integer_correlation_of_the_q8_taps(oracle):
    """cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) (src/ORBextractor.cc:1094-1095), OpenCV's 8-bit path: the
    normalised Gaussian taps are rounded to 8 fractional bits ({18,34,49,55,49,34,18}: they sum to 257, not 256), rows and
    columns are correlated in integers and the 16 fractional bits are rounded off once.  scipy.ndimage.correlate1d on
    int64 with "mirror" boundary is an independent implementation of exactly that: every byte must agree; and the result
    stays within the tap-quantisation bound of the real-valued sigma-2 Gaussian (scipy.ndimage.gaussian_filter)."""
    from scipy import ndimage
    from orb_slam2_ssd_semantic_amd.synth import synth_tum_like
    g = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0)
    taps = np.rint(g / g.sum() * 256).astype(np.int64)
    assert taps.tolist() == [18, 34, 49, 55, 49, 34, 18]
    worst, npx = 0.0, 0
    for seed in range(3):
        for img in (synth_frame(seed, 120, 160), synth_tum_like(seed, 120, 160), synth_frame(seed, 9, 11)):
            rows = ndimage.correlate1d(img.astype(np.int64), taps, axis=1, mode="mirror")
            acc = ndimage.correlate1d(rows, taps, axis=0, mode="mirror")
            theirs = np.minimum((acc + 32768) >> 16, 255).astype(np.uint8)
            ours = oracle.gaussian_blur7(img)[0]
            assert np.array_equal(ours, theirs), seed
            exact = ndimage.gaussian_filter(img.astype(np.float64), 2.0, truncate=1.5, mode="mirror")
            worst = max(worst, float(np.abs(ours.astype(np.float64) - exact).max()))
            npx += img.size
    assert worst < 2.6, worst                   # (257/256 - 1) * 255 + tap rounding + final rounding; a wrong tap is >> 3


def test_bilinear_resize_is_the_rounded_half_pixel_bilinear_of_scipy(oracle):
    """cv::resize(INTER_LINEAR) (src/ORBextractor.cc:1130): source coordinate (x + 0.5) * (src / dst) - 0.5, clamped at the
    edges; scipy.ndimage.map_coordinates(order=1, mode="nearest") evaluates exactly that in float64."""
    from scipy import ndimage
    worst, equal, npx = 0.0, 0, 0
    for seed, (h, w, dh, dw) in enumerate(((480, 640, 400, 533), (400, 533, 333, 444), (96, 128, 80, 107), (50, 70, 42, 58))):
        img = synth_frame(seed, h, w)
        yy = np.clip((np.arange(dh) + 0.5) * (h / dh) - 0.5, 0, h - 1)
        xx = np.clip((np.arange(dw) + 0.5) * (w / dw) - 0.5, 0, w - 1)
        gy, gx = np.meshgrid(yy, xx, indexing="ij")
        exact = ndimage.map_coordinates(img.astype(np.float64), [gy, gx], order=1, mode="nearest")
        ours = oracle.resize_linear(img, dw, dh).astype(np.float64)
        worst = max(worst, float(np.abs(ours - exact).max()))
        equal += int((ours == np.rint(exact)).sum())
        npx += dh * dw
    assert worst < 1.0, worst                   # 11-bit coefficients, truncating >>4 and >>16 stages, final rounding
    assert equal / npx > 0.85, equal / npx     # the rest are off by one grey level (truncating intermediate stages)


def test_fast_atan2_is_within_its_documented_accuracy_of_numpy(oracle):
    """cv::fastAtan2 (src/ORBextractor.cc:87): degrees in [0, 360), documented accuracy about 0.3 degrees."""
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.integers(-200000, 200000, size=(4000, 2)).astype(np.float32),
                        np.array([[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [-1, -1], [1, -1], [3, 1e-3]], np.float32)])
    worst = 0.0
    for y, x in v:
        a = oracle.fast_atan2(float(y), float(x))
        assert 0.0 <= a < 360.0 or a == 360.0
        d = abs(a - (np.degrees(np.arctan2(float(y), float(x))) % 360.0))
        worst = max(worst, min(d, 360.0 - d))
    assert worst < 0.3, worst
    assert oracle.fast_atan2(0.0, 0.0) == 0.0
