"""Randomised parity sweep (GPU box): extractor and brute-force matcher against the oracle on random shapes / parameters.
usage: python tools/fuzz_parity.py [ncases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (same HIP runtime as liborbfe)

from oracle import oracle_ffi as O
from orb_slam2_ssd_semantic_amd import ORBextractor, ORBmatcher, OrbfeError
from orb_slam2_ssd_semantic_amd.synth import synth_frame

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(ncases):
    w, h = int(rng.integers(180, 1000)), int(rng.integers(160, 760))
    nlev = int(rng.integers(1, 9))
    sf = float(np.float32(rng.choice([1.1, 1.2, 1.25, 1.3, 1.4, 1.5, 1.7])))
    nf = int(rng.integers(50, 3000))
    ini = int(rng.integers(8, 40))
    mn = int(rng.integers(1, ini + 1))
    img = synth_frame(int(rng.integers(0, 1 << 30)), h, w, sparse=bool(rng.integers(0, 2)))
    if rng.random() < 0.2:  # flat regions: empty cells exercise the minTh fallback and tiny trees
        y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
        img[y0:y0 + h // 3, x0:x0 + w // 3] = 128
    tag = f"case {c}: {w}x{h} nf={nf} nlev={nlev} sf={sf:.2f} th={ini}/{mn}"
    try:
        e = ORBextractor(nf, sf, nlev, ini, mn, max_width=w, max_height=h)
        gk, gd = e(img)
    except OrbfeError as ex:
        print(tag, "-> rejected:", str(ex)[:90])
        continue
    oe = O.OracleExtractor(nf, sf, nlev, ini, mn)
    ok, od = oe(img, cap=nf + 16 * nlev + 256)
    same = len(gk) == len(ok) and np.array_equal(gd, od) and all(
        np.array_equal(gk[f].view(np.uint32), ok[f].view(np.uint32)) for f in ok.dtype.names)
    if not same:
        bad += 1
        print(tag, "-> MISMATCH", len(gk), len(ok))
    elif c % 10 == 0:
        print(tag, "-> ok", len(gk))
# matcher: random sizes, heavy ties
mt = None
for c in range(ncases):
    nq, nt = int(rng.integers(0, 2500)), int(rng.integers(0, 2500))
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    if nt and nq:
        src = rng.integers(0, nt, nq)
        q = t[src].copy()
        flips = int(rng.integers(0, 70))
        for i in range(nq):
            for b in rng.integers(0, 256, flips):
                q[i, b >> 3] ^= 1 << (b & 7)
        if rng.random() < 0.5:  # duplicate train rows: first index must win, second == best
            t[rng.integers(0, nt, nt // 4)] = t[rng.integers(0, nt, nt // 4)]
    qa = rng.uniform(0, 360, nq).astype(np.float32)
    ta = rng.uniform(0, 360, nt).astype(np.float32)
    ratio, th, ori = float(rng.choice([0.6, 0.75, 0.9])), int(rng.choice([50, 100])), bool(rng.integers(0, 2))
    m = ORBmatcher(ratio, ori)
    got = m.MatchBruteForce(q, t, qa, ta, th)
    ref = O.match_bf(q, t, qa, ta, ratio, th, ori)
    if not (all(np.array_equal(g, r) for g, r in zip(got[:3], ref[:3])) and got[3] == ref[3]):
        bad += 1
        print(f"matcher case {c}: nq={nq} nt={nt} ratio={ratio} th={th} ori={ori} -> MISMATCH")
# the callers either side of the path: generators of the unit tests, fresh seeds
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_grid import grid_case, queries  # noqa: E402
from test_distinctive import make_case  # noqa: E402
from test_bow import make_voc  # noqa: E402
from test_stereo import stereo_pair  # noqa: E402
from orb_slam2_ssd_semantic_amd import FrameGrid, ORBVocabulary  # noqa: E402
mt = ORBmatcher(0.9, True)
for c in range(max(ncases // 4, 5)):
    sd = int(rng.integers(0, 1 << 20))
    xy, octave, minx, miny, gwi, ghi = grid_case(sd, int(rng.integers(0, 3000)))
    g = FrameGrid(mt, xy, octave, minx, miny, gwi, ghi)
    off, idx = O.assign_grid(xy, minx, miny, gwi, ghi)
    okk = np.array_equal(g.cell_off, off) and np.array_equal(g.cell_idx, idx)
    q, lv = queries(sd, int(rng.integers(0, 400)))
    qoff, cand = g.query(q, lv)
    for i in range(len(q)):
        ref = O.features_in_area(xy, octave, off, idx, minx, miny, gwi, ghi, float(q[i, 0]), float(q[i, 1]), float(q[i, 2]),
                                 int(lv[i, 0]), int(lv[i, 1]))
        okk = okk and np.array_equal(cand[qoff[i]:qoff[i + 1]], ref)
    pool, doff, didx = make_case(sd, int(rng.integers(0, 400)), int(rng.integers(1, 90)))
    b, m = mt.ComputeDistinctiveDescriptors(pool, doff, didx)
    rb, rm = O.distinctive(pool, doff, didx)
    okk = okk and np.array_equal(b, rb) and np.array_equal(m, rm)
    voc = make_voc(sd, int(rng.integers(2, 11)), int(rng.integers(1, 5)))
    desc = rng.integers(0, 256, (int(rng.integers(0, 3000)), 32), dtype=np.uint8)
    lu = int(rng.integers(0, 5))
    r = O.bow_transform(voc, desc, lu)
    (bid, bval), (fvn, fvo, fvi) = ORBVocabulary(mt, **voc).transform(desc, lu)
    okk = okk and np.array_equal(bid, r["bow_id"]) and np.array_equal(bval.view(np.uint64), r["bow_val"].view(np.uint64))
    okk = okk and np.array_equal(fvn, r["fv_node"]) and np.array_equal(fvo, r["fv_off"]) and np.array_equal(fvi, r["fv_idx"])
    if c < 4:
        left, right = stereo_pair(sd % 1000)
        exL, exR = O.OracleExtractor(), O.OracleExtractor()
        kL, dL = exL(left)
        kR, dR = exR(right)
        mbf, mb = float(rng.uniform(20, 400)), float(rng.uniform(0.05, 2.0))
        ru, rd, _ = O.stereo_matches(exL, exR, kL, dL, kR, dR, mbf, mb)
        gl = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
        gr = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480)
        gkL, gdL = gl(left)
        gkR, gdR = gr(right)
        u, d = mt.ComputeStereoMatches(gl, gr, gkL, gdL, gkR, gdR, mbf, mb)
        okk = okk and np.array_equal(u.view(np.uint32), ru.view(np.uint32)) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    if not okk:
        bad += 1
        print("callers case", c, "seed", sd, "-> MISMATCH")
print("fuzz done:", ncases, "extractor +", ncases, "matcher cases + grid / distinctive / bow / stereo,", bad, "mismatches")
sys.exit(1 if bad else 0)
