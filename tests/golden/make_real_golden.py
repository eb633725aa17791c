#!/usr/bin/env python3
"""Step 2 of 2 of the real-photograph fixture set: what the COMPILED REFERENCE extracts from the images of step 1.

Runs in the build container (python 3.10, needs /root/reference for oracle/_ref/libref_orb.so = the unmodified
src/ORBextractor.cc compiled by oracle/refbuild/Makefile; bump allocator = creation-order quadtree tie-break, canonical cos/sin).
Writes tests/golden/real/golden.npz:

  px/<tag>                       sha256 of the gray frame the extractor is handed (decoded file -> caller's gray conversion);
                                 a JPEG decoder or PNG reader that yields other pixels is caught before anything is compared
  kps/<tag>, desc/<tag>          the reference's full output for (1000 features, blur_rounding 0): 28-byte cv::KeyPoint records
                                 and 32-byte descriptors in the reference's order
  dig/<tag>/<nf>/<blur>          for every (nfeatures in 1000, 2000) x (blur_rounding 0, 1): [count, sha256(kps bytes),
                                 sha256(desc bytes)] (as a string array) -- the other three configurations pinned by digest
  ncand/<tag>                    candidates per level handed to DistributeOctTree (int32[8])
  native: the same for tests/golden/real/native/*.png with the level count that fits the size (text / page: 4 levels)

Run:  python tests/golden/make_real_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_ffi as R                                  # noqa: E402
from orb_slam2_ssd_semantic_amd import photos                     # noqa: E402

NATIVE_LEVELS = {"text": 4, "page": 4}


def sha(b):
    return hashlib.sha256(np.ascontiguousarray(b).tobytes()).hexdigest()


def frames():
    """[(tag, gray, nlevels)]"""
    out = [(t, g, 8) for t, g in photos.vga_gray_frames()]
    out += [("native:" + n, g, NATIVE_LEVELS.get(n, 8)) for n, g in photos.native_images()]
    return out


def main():
    assert R.have_reference(), "needs /root/reference (the build container)"
    R.build()
    out = {"meta/source": np.array("oracle/_ref/libref_orb.so: unmodified /root/reference/src/ORBextractor.cc, bump allocator, canonical trig")}
    nkp = 0
    for tag, g, nlev in frames():
        out[f"px/{tag}"] = np.array(sha(g))
        for nf in (1000, 2000):
            for blur in (0, 1):
                R.configure(bump=True, canonical_trig=True, blur_mode=blur)
                ref = R.RefExtractor(nf, 1.2, nlev, 20, 7)
                k, d = ref(g, cap=nf + 256)
                out[f"dig/{tag}/{nf}/{blur}"] = np.array([str(len(k)), sha(k.view(np.uint8)), sha(d)])
                if nf == 1000 and blur == 0:
                    out[f"kps/{tag}"] = k.view(np.uint8).reshape(len(k), 28).copy()
                    out[f"desc/{tag}"] = d
                    out[f"ncand/{tag}"] = np.array([len(ref.candidates(l)) for l in range(nlev)], np.int32)
                    nkp += len(k)
    R.configure(bump=True, canonical_trig=True, blur_mode=0)
    path = os.path.join(HERE, "real", "golden.npz")
    np.savez_compressed(path, **out)
    print(f"{len(frames())} frames, {nkp} keypoints at 1000 features; {os.path.getsize(path) / 1e6:.2f} MB -> {path}")


if __name__ == "__main__":
    main()
