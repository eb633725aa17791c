"""`blur_rounding = 1` on the GPU (k_blur7<1>): the rounding a real x86-64 OpenCV <= 3.3 build performs in GaussianBlur's
column pass (SURVEY 9.4 ambiguity A).  SymmColumnVec_32s8u evaluates the column sum in fp32 (exact here) and converts with
cvtps2dq = round-half-to-EVEN for the columns x < (width & ~3); the scalar tail (width % 4 columns) rounds half-UP like
mode 0.  The two modes differ only where the 16-bit fraction of the column sum is exactly 0x8000 AND the half-up value is
odd AND the column is in the vectorised part -- about one pixel in 131 072, so the frames here are CONSTRUCTED to contain
exact halves at chosen places: inside and outside the vectorised part, odd and even, on the image borders.

HIP (through the C-ABI) vs the oracle in the same mode, vs oracle/_ref (the unmodified reference ORBextractor.cc compiled
with the cv stub's GaussianBlur in mode 1), and vs an independent numpy twin (tests/twins.py)."""
import numpy as np
import pytest

import twins
from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like

pytestmark = pytest.mark.gpu

K7 = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)


def _refl(i, n):
    return -i if i < 0 else (2 * (n - 1) - i if i >= n else i)


def column_sum_at(img, x, y):
    """the 32-bit column-pass sum of GaussianBlur at pixel (x, y): sum_j k_j sum_i k_i img[refl(y+j)][refl(x+i)]"""
    h, w = img.shape
    acc = 0
    for j in range(-3, 4):
        row = img[_refl(y + j, h)]
        acc += int(K7[j + 3]) * sum(int(K7[i + 3]) * int(row[_refl(x + i, w)]) for i in range(-3, 4))
    return acc


def plant_exact_half(img, x, y, odd, rng):
    """Rewrite three pixels next to (x, y) so that its column sum has the fraction 0x8000 exactly and the half-up value
    (sum + 32768) >> 16 has the requested parity.  Effective tap weights (reflection folds taps onto one pixel at the
    borders) are measured, not assumed."""
    h, w = img.shape
    xn, yn = (x - 1 if x > 0 else x + 1), (y - 1 if y > 0 else y + 1)
    cand = [(x, y), (xn, y), (xn, yn)]            # three different tap weights (55*55, 49*55, 49*49 away from the borders)
    for _ in range(64):
        base = img.copy()
        for (px, py) in cand:
            base[py, px] = 0
        a0 = column_sum_at(base, x, y)
        wts = []
        for (px, py) in cand:
            t = base.copy()
            t[py, px] = 1
            wts.append(column_sum_at(t, x, y) - a0)
        A = np.arange(256, dtype=np.int64)
        grid = a0 + wts[0] * A[:, None] + wts[1] * A[None, :]
        for c in rng.permutation(256):
            acc = grid + wts[2] * int(c)
            v = (acc + 32768) >> 16
            hit = ((acc & 0xFFFF) == 0x8000) & ((v & 1) == int(odd)) & (v <= 254)
            if hit.any():
                a, b = np.argwhere(hit)[0]
                for (px, py), val in zip(cand, (a, b, c)):
                    img[py, px] = val
                assert column_sum_at(img, x, y) & 0xFFFF == 0x8000
                return
        # no solution with this neighbourhood: perturb a fourth pixel of the window and retry
        img[min(y + 2, h - 1), min(x + 2, w - 1)] = rng.integers(0, 256)
    raise AssertionError("could not plant an exact half")


def frame_with_halves(seed, w, h):
    """S_tum texture with exact halves planted at: vectorised part (odd, even), both tail columns where the width has a
    tail, the left and right borders, the top and bottom rows.  Returns the frame and the list (x, y, odd)."""
    rng = np.random.default_rng(seed)
    img = synth_tum_like(seed, h, w).copy()
    vec_w = w & ~3
    spots = [(100, 50, 1), (203, 61, 0), (0, 120, 1), (vec_w - 1, 140, 1), (317, 0, 1), (322, h - 1, 1), (411, 222, 1)]
    for k, x in enumerate(range(vec_w, w)):             # the scalar tail: w % 4 columns
        spots += [(x, 170 + 20 * k, 1), (x, 300 + 20 * k, 0)]
    if w == vec_w:
        spots += [(w - 1, 170, 1)]                        # the right border inside the vectorised part
    for (x, y, odd) in spots:
        plant_exact_half(img, x, y, odd, rng)
    for (x, y, odd) in spots:                             # plants are 8+ pixels apart: none disturbed another
        acc = column_sum_at(img, x, y)
        assert acc & 0xFFFF == 0x8000 and ((acc + 32768) >> 16) & 1 == odd, (x, y)
    return img, spots


def _same(gk, gd, ok, od):
    return (len(gk) == len(ok) and np.array_equal(gd, od)
            and all(np.array_equal(gk[f].view(np.uint32), ok[f].view(np.uint32)) for f in ok.dtype.names))


@pytest.mark.parametrize("w", [640, 641, 642, 643])
def test_constructed_halves_inside_and_outside_the_vectorised_part(oracle, w):
    from orb_slam2_ssd_semantic_amd import ORBextractor
    h = 480
    img, spots = frame_with_halves(w, w, h)
    vec_w = w & ~3
    ext = {m: ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, blur_rounding=m) for m in (0, 1)}
    out = {}
    for m in (0, 1):
        oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
        oe.set_blur_mode(m)
        ok, od = oe(img)
        gk, gd = ext[m](img)
        for l in range(8):
            assert np.array_equal(ext[m].blurred_level(l), oe.blurred(l)), (m, l)
        assert _same(gk, gd, ok, od), m
        tw, _ = twins.gaussian_blur7(img, sse2=bool(m))           # independent numpy twin of level 0
        assert np.array_equal(ext[m].blurred_level(0), tw), m
        out[m] = ext[m].blurred_level(0).astype(np.int32)
    # the modes differ at level 0 EXACTLY at the odd halves of the vectorised part (planted + whatever the texture holds)
    diff = np.argwhere(out[0] != out[1])
    planted_vec_odd = {(y, x) for (x, y, odd) in spots if odd and x < vec_w}
    planted_rest = {(y, x) for (x, y, odd) in spots if not (odd and x < vec_w)}
    got = {(int(y), int(x)) for y, x in diff}
    assert planted_vec_odd <= got and not (planted_rest & got), (sorted(got), spots)
    for (y, x) in got:
        assert x < vec_w and out[0][y, x] - out[1][y, x] == 1 and out[1][y, x] % 2 == 0
        assert column_sum_at(img, x, y) & 0xFFFF == 0x8000
    if w != vec_w:
        assert any(x >= vec_w and odd for (x, y, odd) in spots)   # an odd half in the scalar tail existed and stayed half-up


def test_mode1_equals_the_compiled_reference_with_sse2_rounding():
    """oracle/_ref (unmodified src/ORBextractor.cc, cv stub's GaussianBlur in mode 1) vs HIP blur_rounding = 1 through one
    batched device call: frames with constructed halves + S / S_tum frames, 1000 and 2000 features."""
    import torch
    from oracle import ref_ffi as R
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
    if not R.available():
        pytest.skip("oracle/_ref/libref_orb.so not present")
    w, h = 640, 480
    frames = [frame_with_halves(50 + i, w, h)[0] for i in range(4)]
    frames += [synth_frame(950 + i, h, w, sparse=(i % 3 == 1)) if i % 2 else synth_tum_like(950 + i, h, w) for i in range(12)]
    frames = np.stack(frames)
    B = len(frames)
    try:
        R.configure(bump=True, canonical_trig=True, blur_mode=1)
        for nf in (1000, 2000):
            e = ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, blur_rounding=1)
            cap = e.capacity()
            dg = torch.from_numpy(frames).cuda()
            dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
            dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
            dn = torch.zeros(B, dtype=torch.int32, device="cuda")
            e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert e.overflow() == 0
            n, kps, desc = dn.cpu().numpy(), dk.cpu().numpy(), dd.cpu().numpy()
            ref = R.RefExtractor(nf, 1.2, 8, 20, 7)
            for i in range(B):
                rk, rd = ref(frames[i], cap=nf + 128)
                gk = kps[i, :n[i]].copy().view(KP_DTYPE).reshape(-1)
                assert n[i] == len(rk), (nf, i)
                assert np.array_equal(gk.view(np.uint8), rk.view(np.uint8)) and np.array_equal(desc[i, :n[i]], rd), (nf, i)
    finally:
        R.configure(bump=True, canonical_trig=True, blur_mode=0)


def test_natural_halves_over_a_sequence_both_modes(oracle):
    """64 frames (odd sizes, so every level has its own tail width) through the batched device path in BOTH modes against the
    oracle in the same mode; the halves the textures happen to contain (about 14 per 640x480 frame over the 8 levels) are
    counted and the two modes must differ in at least one blurred pixel somewhere in the set."""
    import torch
    from orb_slam2_ssd_semantic_amd import KP_DTYPE, ORBextractor
    w, h, B = 637, 479, 64
    frames = np.stack([synth_frame(8800 + i, h, w, sparse=(i % 4 == 1)) if i % 2 else synth_tum_like(8800 + i, h, w) for i in range(B)])
    dg = torch.from_numpy(frames).cuda()
    res, nties, ndiff = {}, 0, 0
    for m in (0, 1):
        e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B, blur_rounding=m)
        cap = e.capacity()
        dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
        dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
        dn = torch.zeros(B, dtype=torch.int32, device="cuda")
        e.extract_batch_device(dg.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert e.overflow() == 0
        n, kps, desc = dn.cpu().numpy(), dk.cpu().numpy(), dd.cpu().numpy()
        oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
        oe.set_blur_mode(m)
        blurred = []
        for i in range(B):
            ok, od = oe(frames[i], cap=1200)
            gk = kps[i, :n[i]].copy().view(KP_DTYPE).reshape(-1)
            assert _same(gk, desc[i, :n[i]], ok, od), (m, i)
            blurred.append([oe.blurred(l).copy() for l in range(8)])
            if m == 0:
                nties += oe.blur_ties()
        res[m] = blurred
    for i in range(B):
        for l in range(8):
            ndiff += int((res[0][i][l] != res[1][i][l]).sum())
    assert nties > 0 and 0 < ndiff <= nties, (nties, ndiff)


def test_blur_rounding_switched_on_an_existing_handle(oracle):
    """ORBFE_OPT_BLUR_ROUNDING: the rounding mode of an existing handle (the plan is rebuilt behind the outstanding work);
    a handle switched 0 -> 1 -> 0 gives what fresh handles of those modes give, on a frame with constructed exact halves
    (where the two modes differ) -- blurred levels, keypoints and descriptors."""
    from orb_slam2_ssd_semantic_amd import ORBextractor
    img = frame_with_halves(91, 640, 480)[0]
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2)
    outs = {}
    for mode in (0, 1, 0):
        e.set_option("blur_rounding", mode)
        gk, gd = e(img)
        oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
        oe.set_blur_mode(mode)
        ok, od = oe(img)
        assert _same(gk, gd, ok, od), mode
        for l in range(8):
            if len(oe.selected(l)):
                assert np.array_equal(e.blurred_level(l), oe.blurred(l)), (mode, l)
        outs.setdefault(mode, []).append(e.blurred_level(0))
    assert not np.array_equal(outs[0][0], outs[1][0])          # the constructed halves make the modes differ
    assert np.array_equal(outs[0][0], outs[0][1])


@pytest.mark.parametrize("w,h", [(640, 480), (533, 401), (322, 243)])
def test_saturating_sums_and_reflected_borders(oracle, w, h):
    """The taps sum to 257, so a window of bright pixels overshoots 255 before saturate_cast<uchar> (:1095, 8-bit GaussianBlur):
    255-blocks give column sums of 257 * 65535 >> 16 = 257 -> 255.  The kernel saturates through the dot product's clamp (an
    accumulator that starts at 0xFF000000 + 2^15), and its border lanes fold BORDER_REFLECT_101 into their weights: saturated
    blocks in the interior, on all four borders and in the corners, 254 / 255 / 0 checkerboards (values around the overshoot), every
    level of three widths (w % 4 = 0, 1, 2), both roundings, against the oracle and -- level 0 -- the numpy twin."""
    from orb_slam2_ssd_semantic_amd import ORBextractor
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[:40, :] = 255                                     # top border rows
    img[-37:, :] = 255                                    # bottom
    img[:, :29] = 255                                     # left columns (reflected taps land on 255s)
    img[:, -31:] = 255                                    # right, incl. the scalar tail of odd widths
    img[60:110, 150:230] = 255                            # interior block
    yy, xx = np.mgrid[0:50, 0:70]
    img[130:180, 60:130] = np.where((yy + xx) & 1, 255, 254).astype(np.uint8)
    img[130:180, 140:210] = np.where((yy // 3 + xx // 3) & 1, 255, 0).astype(np.uint8)
    img[190:205, :] = 253                                 # a band whose blur stays just below / at 255 next to 255s
    for m in (0, 1):
        oe = oracle.OracleExtractor(800, 1.2, 8, 20, 7)
        oe.set_blur_mode(m)
        ok, od = oe(img)
        e = ORBextractor(800, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=16, blur_rounding=m)   # a batch handle: 40-row blur blocks, cell-row FAST
        gk, gd = e(img)
        for l in range(8):
            assert np.array_equal(e.blurred_level(l), oe.blurred(l)), (m, l)
        assert e.blurred_level(0).max() == 255 and _same(gk, gd, ok, od), m
        tw, _ = twins.gaussian_blur7(img, sse2=bool(m))
        assert np.array_equal(e.blurred_level(0), tw), m
        # the batched path (row-block waves of 40 rows, other lane packing) on the same frame
        kb, db = e.extract_batch([img, img[::-1].copy()])[0]
        assert _same(kb, db, ok, od), m
        e.close()
