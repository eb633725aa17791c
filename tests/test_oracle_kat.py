"""Known-answer tests that pin the oracle against tables derivable from the reference source itself
(SURVEY.md 8(c) item 1).  CPU only."""
import hashlib
import math
import struct

import numpy as np
import pytest

import twins


def test_umax_table(oracle):
    # src/ORBextractor.cc:449-465
    assert oracle.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert oracle.umax().tolist() == twins.umax_table()
    assert int(sum(2 * u + 1 for u in oracle.umax()[1:]) * 2 + 31) == 749  # circular patch size


def test_pattern_checksum(oracle):
    p = oracle.pattern().astype(np.int32)
    assert p.shape == (1024,)
    assert int(p.sum()) == -406 and int(p.min()) == -13 and int(p.max()) == 12
    sha = hashlib.sha256(struct.pack("<1024i", *p.tolist())).hexdigest()
    assert sha == "7e645581387b82784797e8adddb9b6f0c12611859fda09ca8a9bec96d767a05f"
    r = np.hypot(p[0::2].astype(float), p[1::2].astype(float)).max()
    assert abs(r - 18.384776310850235) < 1e-12


@pytest.mark.parametrize("nf,expect", [
    (1000, [217, 181, 151, 126, 105, 87, 73, 60]),
    (2000, [434, 362, 302, 251, 209, 175, 145, 122]),
    (4000, [869, 724, 603, 503, 419, 349, 291, 242]),
])
def test_features_per_level(oracle, nf, expect):
    # src/ORBextractor.cc:426-439 (TUM3.yaml: scaleFactor 1.2, 8 levels)
    e = oracle.OracleExtractor(nf, 1.2, 8, 20, 7)
    got = e.features_per_level().tolist()
    assert got == expect and sum(got) == nf


def test_scale_tables(oracle):
    e = oracle.OracleExtractor()
    s, inv, s2, inv2 = e.scales()
    f = np.float32(1.2)
    ref = [np.float32(1.0)]
    for _ in range(7):
        ref.append(np.float32(ref[-1] * f))
    assert s.tolist() == [float(v) for v in ref]
    assert inv.tolist() == [float(np.float32(1.0) / v) for v in ref]
    assert s2.tolist() == [float(np.float32(v * v)) for v in ref]


def test_level_sizes_and_grid(oracle):
    e = oracle.OracleExtractor()
    lw, lh = e.level_sizes(640, 480)
    assert lw.tolist() == [640, 533, 444, 370, 309, 257, 214, 179]
    assert lh.tolist() == [480, 400, 333, 278, 231, 193, 161, 134]
    assert int((lw.astype(np.int64) * lh).sum()) == 950532
    grids = [oracle.cell_grid(int(a), int(b)) for a, b in zip(lw, lh)]
    assert [(g[1], g[2]) for g in grids] == [(20, 14), (16, 12), (13, 10), (11, 8), (9, 6), (7, 5), (6, 4), (4, 3)]
    assert sum(g[1] * g[2] for g in grids) == 815
    lw, lh = e.level_sizes(1920, 1080)
    assert lw.tolist() == [1920, 1600, 1333, 1111, 926, 772, 643, 536]
    assert lh.tolist() == [1080, 900, 750, 625, 521, 434, 362, 301]
    assert int((lw.astype(np.int64) * lh).sum()) == 6419321
    assert sum(g[1] * g[2] for g in (oracle.cell_grid(int(a), int(b)) for a, b in zip(lw, lh))) == 6342
    assert oracle.cell_grid(60, 60)[0] == 0  # too small for one 30-px cell


def test_hamming_is_popcount(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming(a, b) == twins.hamming(a, b)
    z = np.zeros(32, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, ~z) == 256


def test_three_maxima_cases(oracle):
    # src/ORBmatcher.cc:1912-1957 including the two 0.1x rules and tie behaviour (strict >)
    h = [0] * 30
    assert oracle.three_maxima(h) == (-1, -1, -1)
    h[3], h[7], h[9] = 100, 50, 20
    assert oracle.three_maxima(h) == (3, 7, 9)
    h[9] = 9  # max3 < 0.1*max1 -> dropped
    assert oracle.three_maxima(h) == (3, 7, -1)
    h[7] = 9  # max2 < 0.1*max1 -> both dropped
    assert oracle.three_maxima(h) == (3, -1, -1)
    h = [5] * 30  # equal bins: first three indices, nothing dropped
    assert oracle.three_maxima(h) == (0, 1, 2)
    h = [0] * 30
    h[2], h[4] = 10, 1  # 1 < 0.1*10 is false (1.0 < 1.0), second survives; third max3=0 < 1.0 dropped
    assert oracle.three_maxima(h) == (2, 4, -1)
    rng = np.random.default_rng(5)
    for _ in range(300):
        c = rng.integers(0, 40, 30).tolist()
        assert oracle.three_maxima(c) == twins.three_maxima(c)


def test_rot_bin_quirk(oracle):
    # factor = 1/HISTO_LENGTH (sic): bins 0..12 only, round half away from zero (src/ORBmatcher.cc:308-313)
    assert oracle.rot_bin(10.0, 10.0) == 0
    assert oracle.rot_bin(0.0, 1.0) == 12      # 359 / 30 = 11.97 -> 12
    assert oracle.rot_bin(15.0, 0.0) == 1      # 0.5 rounds away from zero
    assert oracle.rot_bin(45.0, 0.0) == 2      # 1.5 -> 2
    rng = np.random.default_rng(3)
    for _ in range(2000):
        a, b = np.float32(rng.uniform(0, 360)), np.float32(rng.uniform(0, 360))
        r = oracle.rot_bin(a, b)
        assert r == twins.rot_bin(a, b) and 0 <= r <= 12


def test_fast_atan2(oracle):
    assert float(oracle.fast_atan2(0, 0)) == 0.0
    assert float(oracle.fast_atan2(0, 1)) == 0.0
    assert abs(float(oracle.fast_atan2(1, 0)) - 90.0) < 1e-4
    assert abs(float(oracle.fast_atan2(0, -1)) - 180.0) < 1e-4
    assert abs(float(oracle.fast_atan2(-1, 0)) - 270.0) < 1e-4
    rng = np.random.default_rng(2)
    for _ in range(3000):
        y, x = rng.integers(-3000000, 3000000, 2)
        got = oracle.fast_atan2(float(y), float(x))
        assert got.view(np.uint32) == twins.fast_atan2(float(y), float(x)).view(np.uint32)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        err = abs(float(got) - ref)
        assert min(err, 360 - err) < 0.02  # documented accuracy of the 7th-order polynomial is ~0.01 deg


def test_canonical_sincos_is_correctly_rounded(oracle):
    # contract: orc_sincos == (float)cos((double)rad), (float)sin((double)rad) -- checked against libm here;
    # the reference's own glibc cosf/sinf may differ in the last bit on a tiny fraction of angles.
    rng = np.random.default_rng(7)
    angles = np.concatenate([np.arange(0, 360, 0.25, dtype=np.float32),
                             rng.uniform(0, 360, 20000).astype(np.float32),
                             np.float32([0, 90, 180, 270, 359.99997, 45, 135, 225, 315])])
    bad = 0
    for a in angles:
        c, s = oracle.sincos(a)
        cr, sr = twins.sincos_correctly_rounded(a)
        bad += int(c.view(np.uint32) != cr.view(np.uint32)) + int(s.view(np.uint32) != sr.view(np.uint32))
    assert bad == 0
