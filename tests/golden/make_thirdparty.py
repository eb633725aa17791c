#!/usr/bin/env python3
"""Generate tests/golden/thirdparty_skimage.npz with a THIRD-PARTY implementation (scikit-image 0.18.3, which is
installed in this container's /opt/conda python 3.9 but not in the interpreter the tests run in, and not on the GPU box):

  * FAST-9/16 segment-test classification (skimage.feature.corner_fast, its own Cython code -- not OpenCV, not ours) on
    seeded images at the two thresholds the extractor uses: an independent anchor for the "is a corner at threshold t"
    half of cv::FAST.  skimage scores corners differently from OpenCV, so only the classification is kept.
  * the 256 rBRIEF test pairs skimage ships (orb_descriptor_positions.txt, copied by its authors from OpenCV's orb.cpp) and
    its intensity-centroid mask / umax table: independent copies of the tables the reference hard-codes
    (src/ORBextractor.cc:135-393, :449-465).

Run with:  /opt/conda/bin/python3.9 tests/golden/make_thirdparty.py
"""
import os

import numpy as np
from skimage.feature import corner_fast, orb

HERE = os.path.dirname(os.path.abspath(__file__))


def image(seed, h=96, w=128):
    """same arithmetic as orb_slam2_ssd_semantic_amd.synth.synth_frame (kept local: this script runs in another interpreter)"""
    rng = np.random.default_rng(seed)
    acc = np.zeros((h, w), np.float64)
    for s, wt in ((4, 0.4), (8, 0.3), (16, 0.2), (32, 0.1)):
        g = rng.integers(0, 256, (-(-h // s), -(-w // s)), dtype=np.uint8)
        acc += wt * np.repeat(np.repeat(g, s, axis=0), s, axis=1)[:h, :w]
    acc += rng.normal(0.0, 2.0, (h, w))
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def main():
    out = {"meta/source": np.array(f"scikit-image {__import__('skimage').__version__} corner_fast(n=9) / orb tables")}
    for seed in (0, 1, 2, 3):
        img = image(seed)
        out[f"fast/img{seed}"] = img
        for t in (7, 20):
            # p brighter than v + t  <=>  p/255 > (v + t + 0.5)/255 for integers: the half keeps float rounding away from ties
            resp = corner_fast(img, n=9, threshold=(t + 0.5) / 255.0)
            out[f"fast/corners{seed}_t{t}"] = np.packbits(resp > 0)
    pos = np.loadtxt(os.path.join(os.path.dirname(orb.__file__), "orb_descriptor_positions.txt"), dtype=np.int8)
    out["orb/positions"] = pos          # 256 x 4: (y0, x0, y1, x1) in skimage's row/col convention
    out["orb/ofast_mask"] = orb.OFAST_MASK.astype(np.uint8)
    out["orb/umax"] = np.array(orb.OFAST_UMAX, np.int32)
    np.savez_compressed(os.path.join(HERE, "thirdparty_skimage.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
