"""The product's link-level matcher shim (orb_slam2_ssd_semantic_amd/shim/ORBmatcher_orbfe.cc) inside the reference's own
class: compiled against the UNMODIFIED /root/reference/include/ORBmatcher.h and linked with the reference's own
ORBmatcher.cc for every other member (oracle/_ref/libshim_ref.so, recipe oracle/refbuild/Makefile).  CPU: it compiles,
links, loads, DescriptorDistance agrees with the reference's.  GPU: both SearchByBoW overloads called through the
reference's class interface return exactly what the reference's compiled bodies return on the same mock objects."""
import numpy as np
import pytest

from oracle import ref_ffi as R
from test_ref_pin import _bow_case

pytestmark = pytest.mark.skipif(not R.shim_available(), reason="oracle/_ref/libshim_ref.so not built and /root/reference absent")


def test_shim_links_against_the_unmodified_reference_header():
    L = R.shim_lib()
    for name in ("shim_descriptor_distance", "shim_search_by_bow_kf_f", "shim_search_by_bow_kf_kf", "shim_three_maxima"):
        assert getattr(L, name)
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    for i in range(0, 200, 2):   # host-side entry point: no GPU needed
        assert R.descriptor_distance(d[i], d[i + 1], shim=True) == R.descriptor_distance(d[i], d[i + 1]) == int(np.unpackbits(d[i] ^ d[i + 1]).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_shim_search_by_bow_equals_reference_bodies(seed):
    rng = np.random.default_rng(300 + seed)
    total = 0
    for it in range(25):
        n1, n2 = int(rng.choice([0, 1, 7, 150, 1000])), int(rng.choice([0, 1, 9, 180, 1000]))
        (d1, v1, a1, fv1), (d2, v2, a2, fv2) = _bow_case(rng, n1, n2, int(rng.choice([1, 4, 30, 120])), 0.8, it % 2)
        nnratio = float(rng.choice([0.6, 0.7, 0.75, 0.9]))
        ori = bool(it % 3)
        rm, rn = R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, nnratio, ori)
        sm, sn = R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, nnratio, ori, shim=True)
        assert sn == rn and np.array_equal(sm, rm), ("KF,F", seed, it, n1, n2)
        r12, rn2 = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, nnratio, ori)
        s12, sn2 = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, nnratio, ori, shim=True)
        assert sn2 == rn2 and np.array_equal(s12, r12), ("KF,KF", seed, it, n1, n2)
        total += rn + rn2
    assert total > 50


def test_extractor_shim_builds_on_its_opencv_path_and_keeps_the_getters():
    """shim/ORBextractor.cc compiled with -DORBFE_WITH_OPENCV against the OpenCV-semantics stub and driven by the
    reference's Frame::ExtractORB (oracle/_ref/libshim_ext.so): it loads, and the getters Frame reads (src/Frame.cc:186-192)
    return the reference constructor's tables before any frame was seen (no GPU involved)."""
    s = R.ShimExtractor(1000, 1.2, 8, 20, 7)
    g = s.getters()
    t = R.RefExtractor(1000, 1.2, 8, 20, 7).tables()
    assert g["levels"] == 8 and abs(g["scale_factor"] - 1.2) < 1e-6
    for a, b in (("scale", "scale"), ("inv_scale", "inv_scale"), ("sigma2", "sigma2"), ("inv_sigma2", "inv_sigma2")):
        assert np.array_equal(g[a].view(np.uint32), t[b].view(np.uint32)), a


@pytest.mark.gpu
def test_reference_frame_extractorb_through_the_shim_equals_the_reference_class():
    """Frame::ExtractORB (the reference's own caller code) -> product shim -> liborbfe.so (HIP)  ==  the reference's own
    ORBextractor class (compiled, bump allocator, canonical sincos): keypoints, descriptors, order; and with mbKeepPyramid
    the public mvImagePyramid (ROI and its 19-px REFLECT_101 border) equals the reference's."""
    from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like
    from test_gpu_blur_rounding import frame_with_halves
    halves = frame_with_halves(77, 640, 480)[0]     # exact halves of GaussianBlur's column sum at known places: the modes differ
    for mode, nf, frames in ((1, 1000, [synth_frame(40), synth_tum_like(41), halves, synth_frame(42, sparse=True)]), (1, 2000, [synth_frame(43)]),
                             (0, 1000, [halves, synth_frame(44)])):
        # the shim's default is the SSE2 column rounding of an x86-64 OpenCV 3.2 build (mnBlurRounding = 1); the reference side
        # is the compiled ORBextractor.cc with the stub's GaussianBlur in the same mode
        R.configure(bump=True, canonical_trig=True, blur_mode=mode)
        shim, ref = R.ShimExtractor(nf, 1.2, 8, 20, 7), R.RefExtractor(nf, 1.2, 8, 20, 7)
        assert shim.blur_rounding() == 1
        shim.set_blur_rounding(mode)
        for i, img in enumerate(frames):
            sk, sd = shim.extract_via_frame(img, left=(i % 2 == 0), keep_pyramid=True, cap=nf + 128)
            rk, rd = ref(img, cap=nf + 128)
            assert len(sk) == len(rk) >= nf
            assert np.array_equal(sk.view(np.uint8), rk.view(np.uint8)) and np.array_equal(sd, rd)
            for l in range(8):
                assert np.array_equal(shim.level(l), ref.level(l)), l
                assert np.array_equal(shim.level(l, with_border=True), ref.level(l, with_border=True)), l
    R.configure(bump=True, canonical_trig=True, blur_mode=0)


@pytest.mark.gpu
def test_reference_stereo_frame_constructor_runs_unchanged_on_the_shims():
    """The reference's stereo Frame constructor (src/Frame.cc:102-168, sliced verbatim; two ExtractORB threads,
    UndistortKeyPoints, ComputeStereoMatches reading the extractors' PUBLIC mvImagePyramid at :649 / :761-778,
    AssignFeaturesToGrid) compiled around the product's extractor shim and matcher shim with NOTHING set on the extractors
    (mbKeepPyramid defaults to the reference's semantics: mvImagePyramid is valid after every operator())  ==  the same
    constructor around the reference's own compiled ORBextractor / ORBmatcher: mvKeys, mvKeysUn, mDescriptors, the right
    image's keypoints, mvuRight and mvDepth bit patterns, the grid, the image bounds."""
    from test_stereo import stereo_pair
    R.configure(bump=True, canonical_trig=True, blur_mode=1)    # the shim's default: an x86-64 OpenCV 3.2 build's column rounding
    L = R.shimstereo_lib()
    ext = (L.shim_st_ext_create(1000, 1.2, 8, 20, 7), L.shim_st_ext_create(1000, 1.2, 8, 20, 7))   # reused across frames, like Tracking's
    matched = 0
    for seed, fx, bf, mb_before in ((5, 500.0, 40.0, 0.0), (6, 718.856, 386.1448, 0.537), (8, 435.2, 47.9, 0.11)):
        left, right = stereo_pair(seed)
        ref = R.stereo_frame(left, right, fx, fx + 1, 319.5, 239.5, bf, 35.0, mb_before=mb_before)
        got = R.stereo_frame(left, right, fx, fx + 1, 319.5, 239.5, bf, 35.0, mb_before=mb_before, shim=True, extractors=ext)
        for k in ("keys", "keys_un", "desc", "keys_right", "desc_right", "cell_off", "cell_idx"):
            assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), (seed, k)
        for k in ("u_right", "depth", "scal"):
            assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (seed, k)
        matched += int((ref["u_right"] >= 0).sum())
    assert matched > 300
    for e in ext:
        L.shim_st_ext_destroy(e)
    R.configure(bump=True, canonical_trig=True, blur_mode=0)


@pytest.mark.gpu
def test_reference_rgbd_mono_and_masked_frame_constructors_run_unchanged_on_the_shim():
    """the callers of the TUM path -- Frame(imGray, imDepth, ...) src/Frame.cc:176-245 with
    ComputeStereoFromRGBD :850-874, the monocular constructor :247-311, perfect/'s constructor with the dynamic-object mask
    perfect/src/Frame.cc:328-420 -- sliced verbatim and compiled around the PRODUCT's extractor shim (libshim_stereo.so), one
    extractor object re-used over the frames like Tracking's  ==  the same compiled constructors around the reference's own
    ORBextractor: mvKeys, mvKeysUn, mvuRight, mvDepth, mDescriptors and mGrid bit patterns, the image bounds.  Synthetic depth
    (with holes) and mask; S frames, a sparse one and camera-like S_tum frames; both blur roundings."""
    from orb_slam2_ssd_semantic_amd.synth import synth_frame, synth_tum_like
    from test_ref_pin import tum_like_depth_and_mask
    fx, fy, cx, cy, bf = 535.4, 539.2, 320.1, 247.6, 40.0
    L = R.shimstereo_lib()
    total = 0
    for mode in (1, 0):   # 1 = the shim's default (what an x86-64 OpenCV 3.2 binary computes)
        R.configure(bump=True, canonical_trig=True, blur_mode=mode)
        ext = L.shim_st_ext_create(1000, 1.2, 8, 20, 7)
        if mode == 0:
            L.shim_st_ext_set_blur_rounding(ext, 0)
        rext = R.RefExtractor(1000, 1.2, 8, 20, 7)
        frames = [synth_frame(500), synth_tum_like(501), synth_frame(502, sparse=True), synth_tum_like(503)]
        for i, gray in enumerate(frames):
            depth, mask = tum_like_depth_and_mask(40 + i, masked_frac=0.5 if i == 3 else 0.25)
            for kind in (R.FRAME_RGBD, R.FRAME_MONO, R.FRAME_MASKED):
                args = (kind, gray, depth if kind != R.FRAME_MONO else None, mask if kind == R.FRAME_MASKED else None, fx, fy, cx, cy, bf, 40.0)
                ref = R.frame_ctor(*args, extractor=rext)
                got = R.frame_ctor(*args, shim=True, extractor=ext)
                assert got["N"] == ref["N"] > 0, (mode, i, kind)
                for k in ("keys", "keys_un", "desc", "cell_off", "cell_idx"):
                    assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), (mode, i, kind, k)
                for k in ("u_right", "depth", "scal"):
                    assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), (mode, i, kind, k)
                total += got["N"]
        L.shim_st_ext_destroy(ext)
    assert total > 15000
    R.configure(bump=True, canonical_trig=True, blur_mode=0)
