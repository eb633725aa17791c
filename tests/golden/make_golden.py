#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

The reference cannot be built or imported anywhere we run (no OpenCV / DBoW2), so these
vectors come from the CPU ORACLE (oracle/orb_oracle.c), which is itself pinned by the known-answer and
definition-level-twin tests.  They freeze the oracle's outputs so that (a) an accidental change of the
oracle is caught on CPU and (b) the GPU path can be checked on the GPU box without re-deriving them.
If a machine with OpenCV 3.2 becomes available, a dump from the unmodified reference replaces this file's
output (same keys).  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_ffi as O  # noqa: E402
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
    out = {}
    for name, seed, h, w, sparse, nf in CASES:
        img = synth_frame(seed, h, w, sparse)
        e = O.OracleExtractor(nf, 1.2, 8, 20, 7)
        kps, desc = e(img)
        out[f"{name}/kps"] = kps
        out[f"{name}/desc"] = desc
        out[f"{name}/img_sha"] = np.array(sha(img))
        out[f"{name}/level_sha"] = np.array([sha(e.level(l)) for l in range(8)])
        out[f"{name}/blur_sha"] = np.array([sha(e.blurred(l)) if e.blurred(l) is not None else "" for l in range(8)])
        out[f"{name}/ncand"] = np.array([len(e.candidates(l)) for l in range(8)], np.int32)
        out[f"{name}/cand_sha"] = np.array([sha(e.candidates(l)) for l in range(8)])
        out[f"{name}/nsel"] = np.array([len(e.selected(l)) for l in range(8)], np.int32)
        print(name, len(kps), out[f"{name}/ncand"].tolist())
    # matcher golden: frame 0 vs frame 1 descriptors, config-3 parameters
    k0, d0 = out["A_dense_s0/kps"], out["A_dense_s0/desc"]
    k1, d1 = out["A_dense_s1/kps"], out["A_dense_s1/desc"]
    m, b, s, n = O.match_bf(d1, d0, k1["angle"], k0["angle"], 0.9, 100, True)
    out["bf_1to0/match"], out["bf_1to0/best"], out["bf_1to0/second"] = m, b, s
    out["bf_1to0/n"] = np.array(n, np.int32)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb_golden.npz"), **out)


if __name__ == "__main__":
    main()
