#!/usr/bin/env python3
"""Generate tests/golden/orb_golden.npz FROM THE COMPILED REFERENCE (oracle/_ref/libref_orb.so).

libref_orb.so is the unmodified /root/reference/src/ORBextractor.cc and src/ORBmatcher.cc built by
oracle/refbuild/Makefile (cv stub for the types; the five OpenCV algorithms -- resize, copyMakeBorder, FAST,
GaussianBlur, fastAtan2 -- are the oracle's restatements of OpenCV 3.2, so for those five stages these vectors
are not independent of the oracle).  Settings: bump allocator (the :686 pointer sort then breaks ties by
creation order), canonical cos/sin at :97, integer half-up blur rounding.  /root/reference exists only in the
build container, so this script runs there and the vectors are committed; tests/test_golden_cpu.py checks
the oracle against them on CPU and tests/test_gpu_extract.py checks the HIP path against them on the GPU box.
The "bf_1to0" block is the config-3 brute-force matcher, which is NOT a reference function:
those vectors are the oracle's.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O  # noqa: E402
from oracle import ref_ffi as R  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frame  # noqa: E402

CASES = [
    # name, seed, h, w, sparse, nfeatures
    ("A_dense_s0", 0, 480, 640, False, 1000),
    ("A_dense_s1", 1, 480, 640, False, 1000),
    ("A_sparse_s2", 2, 480, 640, True, 1000),
    ("B_dense_s10000", 10000, 480, 640, False, 2000),
    ("odd_517x389_s7", 7, 389, 517, False, 500),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert R.have_reference(), "goldens are generated from the compiled reference: run where /root/reference exists"
    R.configure(bump=True, canonical_trig=True, blur_mode=0)
    out = {"meta/source": np.array("oracle/_ref/libref_orb.so = unmodified reference ORBextractor.cc/ORBmatcher.cc; "
                                   "bump allocator, canonical sincos, blur half-up")}
    for name, seed, h, w, sparse, nf in CASES:
        img = synth_frame(seed, h, w, sparse)
        e = R.RefExtractor(nf, 1.2, 8, 20, 7)
        kps, desc = e(img, cap=nf + 128)
        per_level = e.keypoints_octtree(img, cap=nf + 128)
        kps, desc = e(img, cap=nf + 128)      # again, so the taps below belong to operator()
        out[f"{name}/kps"] = kps
        out[f"{name}/desc"] = desc
        out[f"{name}/img_sha"] = np.array(sha(img))
        out[f"{name}/level_sha"] = np.array([sha(e.level(l)) for l in range(8)])
        blurred = e.blurred()
        bsha, bi = [], 0
        for l in range(8):
            if len(per_level[l]):
                bsha.append(sha(blurred[bi]))
                bi += 1
            else:
                bsha.append("")
        out[f"{name}/blur_sha"] = np.array(bsha)
        out[f"{name}/ncand"] = np.array([len(e.candidates(l)) for l in range(8)], np.int32)
        out[f"{name}/cand_sha"] = np.array([sha(e.candidates(l)) for l in range(8)])
        out[f"{name}/nsel"] = np.array([len(per_level[l]) for l in range(8)], np.int32)
        print(name, len(kps), out[f"{name}/ncand"].tolist())
    # SearchByBoW x2 from the reference matcher: frame 1 against frame 0, nodes = coarse position buckets
    k0, d0 = out["A_dense_s0/kps"], out["A_dense_s0/desc"]
    # second view of the same scene: frame 0 shifted by (+3, +2) px with fresh noise, so descriptors correlate
    img2 = np.roll(synth_frame(0), (2, 3), axis=(0, 1)).astype(np.float64)
    img2 = np.clip(np.rint(img2 + np.random.default_rng(123).normal(0, 3.0, img2.shape)), 0, 255).astype(np.uint8)
    k1, d1 = R.RefExtractor(1000, 1.2, 8, 20, 7)(img2)
    out["bow/k1"], out["bow/d1"] = k1, d1

    def featvec(k):
        node = (k["octave"].astype(np.int64) * 64 + (k["y"] // 80).astype(np.int64) * 8 + (k["x"] // 80).astype(np.int64))
        ids = np.unique(node)
        idx = [np.flatnonzero(node == v) for v in ids]
        off = np.concatenate([[0], np.cumsum([len(i) for i in idx])]).astype(np.uint32)
        return ids.astype(np.uint32), off, np.concatenate(idx).astype(np.uint32)

    fv0, fv1 = featvec(k0), featvec(k1)
    valid0 = (np.arange(len(k0)) % 7 != 3).astype(np.uint8)
    valid1 = (np.arange(len(k1)) % 5 != 1).astype(np.uint8)
    m, n = R.search_by_bow_kf_f(d0, valid0, k0["angle"], fv0, d1, k1["angle"], fv1, 0.7, True)
    out["bow_kf_f/match"], out["bow_kf_f/n"] = m, np.array(n, np.int32)
    m, n = R.search_by_bow_kf_kf(d0, valid0, k0["angle"], fv0, d1, valid1, k1["angle"], fv1, 0.75, True)
    out["bow_kf_kf/match12"], out["bow_kf_kf/n"] = m, np.array(n, np.int32)
    for i, fv in enumerate((fv0, fv1)):
        out[f"bow/fv{i}_node"], out[f"bow/fv{i}_off"], out[f"bow/fv{i}_idx"] = fv
    out["bow/valid0"], out["bow/valid1"] = valid0, valid1
    # brute-force matcher (config 3; not a reference function -> oracle output): frame 1 vs frame 0
    k1, d1 = out["A_dense_s1/kps"], out["A_dense_s1/desc"]
    m, b, s, n = O.match_bf(d1, d0, k1["angle"], k0["angle"], 0.9, 100, True)
    out["bf_1to0/match"], out["bf_1to0/best"], out["bf_1to0/second"] = m, b, s
    out["bf_1to0/n"] = np.array(n, np.int32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb_golden.npz"), **out)


if __name__ == "__main__":
    main()
