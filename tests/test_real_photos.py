"""Real photographs (tests/golden/real/) through the CPU side: the oracle against golden vectors made by the COMPILED REFERENCE.

TUM fr3/walking_xyz is on no box; these are real camera images from the build container (README.md in that directory).
golden.npz was written by tests/golden/make_real_golden.py from oracle/_ref/libref_orb.so (the unmodified src/ORBextractor.cc).
Checked here, without a GPU: (1) the image files decode to the pixels the goldens were made from (PNG reader, JPEG IDCT, the
caller's gray conversion src/Tracking.cc:339-353); (2) the oracle reproduces the reference's output on every frame: the full
keypoint records / descriptors / order for (1000 features, blur_rounding 0), counts + sha256 for 2000 features and the SSE2 blur
rounding; candidates per level; (3) where the reference sources are present, the freshly compiled reference still equals its goldens.
"""
import hashlib
import os

import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd import KP_DTYPE, photos

GOLD = os.path.join(photos.ROOT, "golden.npz")
NATIVE_LEVELS = {"text": 4, "page": 4}


def sha(b):
    return hashlib.sha256(np.ascontiguousarray(b).tobytes()).hexdigest()


def all_frames():
    out = [(t, g, 8) for t, g in photos.vga_gray_frames()]
    out += [("native:" + n, g, NATIVE_LEVELS.get(n, 8)) for n, g in photos.native_images()]
    return out


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def frames():
    return all_frames()


def test_fixture_files_decode_to_the_pixels_the_goldens_were_made_from(gold, frames):
    assert len(frames) == 40
    sizes = set()
    for tag, g, _ in frames:
        assert g.dtype == np.uint8 and g.ndim == 2
        assert sha(g) == str(gold[f"px/{tag}"]), f"{tag}: decoder or gray conversion yields other pixels than the fixture generator's"
        sizes.add(g.shape)
    assert (480, 640) in sizes and len(sizes) >= 8      # VGA + seven native sizes
    # the two Camera.RGB settings give different gray frames (R and B weights swap): both are in the set
    d = dict((t, g) for t, g, _ in frames)
    assert not np.array_equal(d["coffee@rgb1"], d["coffee@rgb0"])


def test_oracle_equals_the_compiled_reference_on_real_photographs(oracle, gold, frames):
    nkp = 0
    for tag, g, nlev in frames:
        for nf in (1000, 2000):
            for blur in (0, 1):
                oe = oracle.OracleExtractor(nf, 1.2, nlev, 20, 7)
                oe.set_blur_mode(blur)
                k, d = oe(g, cap=nf + 256)
                n, hk, hd = gold[f"dig/{tag}/{nf}/{blur}"].tolist()
                assert (len(k), sha(k.view(np.uint8)), sha(d)) == (int(n), hk, hd), (tag, nf, blur)
                if nf == 1000 and blur == 0:
                    gk = gold[f"kps/{tag}"].reshape(-1).view(KP_DTYPE)
                    for f in KP_DTYPE.names:
                        assert np.array_equal(k[f].view(np.uint32), gk[f].view(np.uint32)), (tag, f)
                    assert np.array_equal(d, gold[f"desc/{tag}"]), tag
                    assert [len(oe.candidates(l)) for l in range(nlev)] == gold[f"ncand/{tag}"].tolist(), tag
                    nkp += len(k)
    assert nkp > 35000


def test_freshly_compiled_reference_equals_its_goldens(gold, frames):
    from oracle import ref_ffi as R
    if not R.available():
        pytest.skip("oracle/_ref/libref_orb.so not present (built where /root/reference exists)")
    try:
        for blur in (0, 1):
            R.configure(bump=True, canonical_trig=True, blur_mode=blur)
            for tag, g, nlev in frames[::3]:
                ref = R.RefExtractor(1000, 1.2, nlev, 20, 7)
                k, d = ref(g, cap=1256)
                n, hk, hd = gold[f"dig/{tag}/1000/{blur}"].tolist()
                assert (len(k), sha(k.view(np.uint8)), sha(d)) == (int(n), hk, hd), (tag, blur)
    finally:
        R.configure(bump=True, canonical_trig=True, blur_mode=0)
