"""ctypes binding of oracle/_ref/libref_orb.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

libref_orb.so is the UNMODIFIED /root/reference/src/ORBextractor.cc and src/ORBmatcher.cc compiled by
oracle/refbuild/Makefile against a cv stub (types) whose five OpenCV algorithms are the oracle's restatements.
It exists to pin oracle/orb_oracle.c to code compiled from the reference.  Only tests/ (and, as a CPU baseline,
bench.py's cpu_baseline leg) may import this module; the product package never does.

/root/reference exists only in the build container: there build() runs the Makefile; on the GPU box the
prebuilt library travels with the snapshot and build() only checks that it is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_OUT = os.path.join(_HERE, "_ref")
_LIB = os.path.join(_OUT, "libref_orb.so")
REFERENCE = os.environ.get("ORBFE_REFERENCE", "/root/reference")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4")])


def have_reference():
    return os.path.exists(os.path.join(REFERENCE, "src", "ORBextractor.cc"))


def available():
    return os.path.exists(_LIB) or have_reference()


def _make(target):
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "refbuild"), f"REF={REFERENCE}", target])


def build():
    """make -C oracle/refbuild ref (libref_orb.so only: gcc/g++ + the reference sources, nothing of the product) when the
    reference sources are present; otherwise require the prebuilt library."""
    if have_reference():
        _make("ref")
    if not os.path.exists(_LIB):
        raise RuntimeError("oracle/_ref/libref_orb.so is missing and /root/reference is not available to build it")
    return _LIB


def build_shims():
    """make -C oracle/refbuild shims: the product's shims inside the reference's classes.  Needs the product library
    (orb_slam2_ssd_semantic_amd/liborbfe.so, hipcc) to exist; called only by the shim tests / the driver's build()."""
    build()
    if have_reference():
        import sys
        root = os.path.dirname(_HERE)
        if root not in sys.path:
            sys.path.insert(0, root)
        from orb_slam2_ssd_semantic_amd import _build
        _build.build()  # hipcc; raises if the product library cannot be built
        _make("shims")


_lib = None
_PERFECT = os.path.join(_OUT, "libref_perfect.so")
_perfect = None


class RefFrameArgs(C.Structure):
    """oracle/refbuild/ref_matcher_api.cpp: the current frame of the projection searches as flat arrays"""
    _fields_ = [("desc", C.c_void_p), ("xy", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p),
                ("uRight", C.c_void_p), ("state", C.c_void_p), ("n", C.c_int), ("Tcw", C.c_void_p),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("mb", C.c_float), ("minx", C.c_float), ("maxx", C.c_float), ("miny", C.c_float), ("maxy", C.c_float),
                ("gw_inv", C.c_float), ("gh_inv", C.c_float), ("scale_factors", C.c_void_p), ("nlevels", C.c_int)]


def perfect_lib():
    """oracle/_ref/libref_perfect.so: the same recipe over perfect/src + perfect/include (the reference's second copy of
    the path; its ORBmatcher has the extra SearchByProjection overload of SURVEY 8(a) M9)"""
    global _perfect
    if _perfect is None:
        build()
        if not os.path.exists(_PERFECT):
            raise RuntimeError("oracle/_ref/libref_perfect.so is missing")
        _perfect = _declare(C.CDLL(_PERFECT))
    return _perfect


_VEC = os.path.join(_OUT, "libref_orb_vec.so")
_vec = None


def vectorised_available():
    """oracle/_ref/libref_orb_vec.so exists and this host's CPUs have AVX2 (it is built -O3 -mavx2)"""
    if not os.path.exists(_VEC):
        return False
    try:
        return " avx2" in open("/proc/cpuinfo").read()
    except OSError:
        return False


def vectorised_lib():
    global _vec
    if _vec is None:
        if not vectorised_available():
            raise RuntimeError("oracle/_ref/libref_orb_vec.so is missing or this CPU has no AVX2")
        _vec = _declare(C.CDLL(_VEC))
    return _vec


class use_vectorised:
    """with use_vectorised(): every ref_* call goes to libref_orb_vec.so -- the same unmodified reference sources and stub
    stand-ins as libref_orb.so, built -O3 -mavx2 -ffp-contract=off (bit-identical results, tests/test_ref_pin.py)"""

    def __enter__(self):
        global _lib
        self.prev = lib()
        _lib = vectorised_lib()
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    _lib = _declare(C.CDLL(_LIB))
    return _lib


class use_perfect:
    """with use_perfect(): every ref_* call of this module (RefExtractor, search_by_bow_*, ...) goes to
    libref_perfect.so -- the reference's perfect/src + perfect/include copy -- instead of libref_orb.so"""

    def __enter__(self):
        global _lib
        self.prev = lib()
        _lib = perfect_lib()
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def _declare(L):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.ref_ext_create.restype = vp
    L.ref_ext_create.argtypes = [ci, cf, ci, ci, ci]
    L.ref_ext_destroy.argtypes = [vp]
    L.ref_ext_tables.argtypes = [vp] * 8
    L.ref_ext_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, vp]
    L.ref_ext_keypoints_octtree.argtypes = [vp, vp, ci, ci, ci, vp, ci, vp]
    L.ref_ext_level.argtypes = [vp, ci, ci, vp, ci, vp, vp]
    L.ref_ext_num_candidates.argtypes = [vp, ci]
    L.ref_ext_candidates.argtypes = [vp, ci, vp]
    L.ref_ext_num_blur.argtypes = [vp]
    L.ref_ext_blur.argtypes = [vp, ci, vp, ci, vp, vp]
    L.ref_ext_fast_calls.restype = C.c_long
    L.ref_ext_fast_calls.argtypes = [vp]
    L.ref_ext_blur_ties.restype = C.c_long
    L.ref_ext_blur_ties.argtypes = [vp]
    L.ref_distribute_octtree.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, ci]
    L.ref_config_bump.argtypes = [ci]
    L.ref_set_trig_mode.argtypes = [ci]
    L.ref_set_blur_mode.argtypes = [ci]
    L.ref_glibc_sincosf.argtypes = [cf, vp, vp]
    L.ref_descriptor_distance.argtypes = [vp, vp]
    L.ref_three_maxima.argtypes = [vp, ci, vp, vp, vp]
    L.ref_search_by_bow_kf_f.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, cf, ci, vp]
    L.ref_search_by_bow_kf_kf.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci] * 2 + [cf, ci, vp]
    L.ref_matcher_constants.argtypes = [vp, vp, vp]
    L.ref_assign_grid.argtypes = [vp, ci, cf, cf, cf, cf, vp, vp]
    L.ref_features_in_area.argtypes = [vp, vp, ci, vp, vp, cf, cf, cf, cf, cf, cf, cf, ci, ci, vp, ci]
    L.ref_distinctive.argtypes = [vp, ci, vp, vp, ci, vp, vp]
    L.ref_predict_scale.argtypes = [cf, cf, cf, ci]
    L.ref_stereo_matches.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp]
    _declare_projection(L, "ref_")
    return L


def _declare_projection(L, prefix):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    getattr(L, prefix + "search_by_projection_last_frame").argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, cf, ci,
                                                                       vp, vp, vp, vp]
    getattr(L, prefix + "search_by_projection_local_map").argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, cf, vp]
    getattr(L, prefix + "last_call_ms").restype = C.c_double
    getattr(L, prefix + "search_by_projection_frame_kf").argtypes = [vp, cf, ci, vp, vp, vp, vp, vp, vp, vp, vp, cf, ci, ci, vp]
    getattr(L, prefix + "search_by_projection_kf_sim3").argtypes = [vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp]
    getattr(L, prefix + "search_for_triangulation").argtypes = [vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp, ci, vp]
    getattr(L, prefix + "search_for_initialization").argtypes = [vp, vp, vp, ci, cf, ci, vp]
    getattr(L, prefix + "search_by_sim3").argtypes = [vp, vp, vp, vp, cf, vp, vp, cf, vp]
    getattr(L, prefix + "fuse_sim3").argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, cf, vp, vp]
    getattr(L, prefix + "fuse").argtypes = ([vp] * 6 + [ci, vp, vp, vp] + [cf] * 11 + [vp, vp, ci, cf, ci] + [vp] * 9 + [cf, vp, vp, vp])


_SHIM = os.path.join(_OUT, "libshim_ref.so")
_SHIM_PERFECT = os.path.join(_OUT, "libshim_perfect.so")
_shim = None
_shim_perfect = None


def shim_available():
    return os.path.exists(_SHIM) or have_reference()


def shim_lib():
    """oracle/_ref/libshim_ref.so: the PRODUCT's link-level matcher shim (shim/ORBmatcher_orbfe.cc) compiled against the
    reference's unmodified include/ORBmatcher.h and linked with the reference's own ORBmatcher.cc for every other member.
    Same flat entry points as the ref_* ones, exported as shim_*; the SearchByBoW calls need a GPU."""
    global _shim
    if _shim is not None:
        return _shim
    build_shims()
    if not os.path.exists(_SHIM):
        raise RuntimeError("oracle/_ref/libshim_ref.so is missing")
    L = C.CDLL(_SHIM)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.shim_descriptor_distance.argtypes = [vp, vp]
    L.shim_three_maxima.argtypes = [vp, ci, vp, vp, vp]
    L.shim_search_by_bow_kf_f.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, cf, ci, vp]
    L.shim_search_by_bow_kf_kf.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci] * 2 + [cf, ci, vp]
    L.shim_matcher_constants.argtypes = [vp, vp, vp]
    _declare_projection(L, "shim_")
    _shim = L
    return L


_SHIM_FULL = os.path.join(_OUT, "libshim_full.so")
_shim_full = None


def shim_full_lib():
    """oracle/_ref/libshim_full.so: shim/ORBmatcher_orbfe.cc with -DORBFE_SHIM_STANDALONE as the ONLY ORBmatcher translation
    unit (the reference's src/ORBmatcher.cc is not in this library); same shim_* entry points.  Pass shim="full"."""
    global _shim_full
    if _shim_full is None:
        build_shims()
        if not os.path.exists(_SHIM_FULL):
            raise RuntimeError("oracle/_ref/libshim_full.so is missing")
        L = C.CDLL(_SHIM_FULL)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.shim_descriptor_distance.argtypes = [vp, vp]
        L.shim_three_maxima.argtypes = [vp, ci, vp, vp, vp]
        L.shim_search_by_bow_kf_f.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, cf, ci, vp]
        L.shim_search_by_bow_kf_kf.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci] * 2 + [cf, ci, vp]
        L.shim_matcher_constants.argtypes = [vp, vp, vp]
        _declare_projection(L, "shim_")
        _shim_full = L
    return _shim_full


def shim_perfect_lib():
    """oracle/_ref/libshim_perfect.so: the same shim compiled with -DORBFE_SHIM_PERFECT inside perfect/'s ORBmatcher class
    (the overload of SearchByProjection that also returns the 2-D point pairs, perfect/src/ORBmatcher.cc:1727-1911)"""
    global _shim_perfect
    if _shim_perfect is None:
        build_shims()
        if not os.path.exists(_SHIM_PERFECT):
            raise RuntimeError("oracle/_ref/libshim_perfect.so is missing")
        L = C.CDLL(_SHIM_PERFECT)
        _declare_projection(L, "shim_")
        _shim_perfect = L
    return _shim_perfect


_SHIMEXT = os.path.join(_OUT, "libshim_ext.so")
_shimext = None


def shimext_lib():
    """oracle/_ref/libshim_ext.so: the PRODUCT's extractor shim (shim/ORBextractor.cc on its -DORBFE_WITH_OPENCV path)
    driven by the reference's own Frame::ExtractORB (sliced from src/Frame.cc:337-343).  Extraction needs a GPU."""
    global _shimext
    if _shimext is not None:
        return _shimext
    build_shims()
    if not os.path.exists(_SHIMEXT):
        raise RuntimeError("oracle/_ref/libshim_ext.so is missing")
    L = C.CDLL(_SHIMEXT)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.shimext_create.restype = vp
    L.shimext_create.argtypes = [ci, cf, ci, ci, ci]
    L.shimext_destroy.argtypes = [vp]
    L.shimext_set_blur_rounding.argtypes = [vp, ci]
    L.shimext_get_blur_rounding.argtypes = [vp]
    L.shimext_extract_via_frame.argtypes = [vp, ci, vp, ci, ci, ci, vp, vp, ci, ci]
    L.shimext_level.argtypes = [vp, ci, ci, vp, ci, vp, vp]
    L.shimext_getters.argtypes = [vp] * 7
    _shimext = L
    return L


class ShimExtractor:
    """ORB_SLAM2::ORBextractor of the product's shim, called the way the reference's Frame calls it."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = shimext_lib()
        self.h = self.L.shimext_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.shimext_destroy(self.h)
            self.h = None

    def set_blur_rounding(self, mode):
        self.L.shimext_set_blur_rounding(self.h, int(mode))

    def blur_rounding(self):
        return self.L.shimext_get_blur_rounding(self.h)

    def set_reuse(self, on):
        """mbReuseIdenticalInput of the shim (before the first call)"""
        self.L.shimext_set_reuse.argtypes = [C.c_void_p, C.c_int]
        self.L.shimext_set_reuse(self.h, int(on))

    def reused_calls(self):
        self.L.shimext_reused_calls.argtypes = [C.c_void_p]
        self.L.shimext_reused_calls.restype = C.c_long
        return int(self.L.shimext_reused_calls(self.h))

    def getters(self):
        n = self.nlevels
        lv, sf = C.c_int(), C.c_float()
        a, b, c, d = (np.zeros(n, np.float32) for _ in range(4))
        self.L.shimext_getters(self.h, C.byref(lv), C.byref(sf), _p(a), _p(b), _p(c), _p(d))
        return dict(levels=lv.value, scale_factor=sf.value, scale=a, inv_scale=b, sigma2=c, inv_sigma2=d)

    def extract_via_frame(self, image, left=True, keep_pyramid=False, cap=None):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.shimext_extract_via_frame(self.h, int(left), _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap,
                                             int(keep_pyramid))
        if n < 0:
            raise RuntimeError(f"shimext_extract_via_frame rc={n}")
        return kps[:n].copy(), desc[:n].copy()

    def level(self, level, with_border=False, cap=1 << 24):
        buf = np.zeros(cap, np.uint8)
        w, h = C.c_int(), C.c_int()
        rc = self.L.shimext_level(self.h, level, int(with_border), _p(buf), cap, C.byref(w), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"shimext_level rc={rc}")
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


_SHIMSTEREO = os.path.join(_OUT, "libshim_stereo.so")
_shimstereo = None
_STEREO_ARGS = None


def _declare_stereo_frame(fn):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fn.argtypes = [vp, vp, vp, vp, ci, ci, ci] + [cf] * 7 + [vp] * 7 + [ci, vp, vp, vp, vp]
    fn.restype = ci


def shimstereo_lib():
    """oracle/_ref/libshim_stereo.so: the reference's stereo Frame constructor (src/Frame.cc:102-168, sliced verbatim with
    ExtractORB / UndistortKeyPoints / ComputeImageBounds / ComputeStereoMatches / AssignFeaturesToGrid) compiled around the
    PRODUCT's extractor shim and matcher shim.  Needs a GPU to run."""
    global _shimstereo
    if _shimstereo is None:
        build_shims()
        if not os.path.exists(_SHIMSTEREO):
            raise RuntimeError("oracle/_ref/libshim_stereo.so is missing")
        L = C.CDLL(_SHIMSTEREO)
        L.shim_st_ext_create.restype = C.c_void_p
        L.shim_st_ext_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.shim_st_ext_destroy.argtypes = [C.c_void_p]
        L.shim_st_ext_set_blur_rounding.argtypes = [C.c_void_p, C.c_int]
        L.shim_st_ext_set_reuse.argtypes = [C.c_void_p, C.c_int]
        L.shim_st_ext_set_keep_pyramid.argtypes = [C.c_void_p, C.c_int]
        L.shim_st_ext_reused_calls.argtypes = [C.c_void_p]
        L.shim_st_ext_reused_calls.restype = C.c_long
        _declare_stereo_frame(L.shim_st_stereo_frame)
        _shimstereo = L
    return _shimstereo


def stereo_frame(left, right, fx, fy, cx, cy, bf, th_depth, mb_before=0.0, nfeatures=1000, scale_factor=1.2, nlevels=8,
                 ini_th=20, min_th=7, shim=False, extractors=None):
    """Frame(imLeft, imRight, timeStamp, extractorLeft, extractorRight, voc, K, distCoef = 0, bf, thDepth) -- the reference's
    stereo constructor body, compiled -- around two reference extractors (shim=False, libref_orb.so) or two of the
    product's shim extractors + the product's matcher shim (shim=True, libshim_stereo.so).  Returns a dict of what the
    constructor leaves in the Frame."""
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    cap = nfeatures + 4 * nlevels + 256
    keys, keys_un, keys_r = (np.zeros(cap, KP_DTYPE) for _ in range(3))
    desc, desc_r = np.zeros((cap, 32), np.uint8), np.zeros((cap, 32), np.uint8)
    ur, dep = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    cell_off, cell_idx = np.zeros(64 * 48 + 1, np.uint32), np.zeros(cap, np.uint32)
    scal = np.zeros(8, np.float32)
    nr = C.c_int()
    if shim:
        L = shimstereo_lib()
        made = extractors is None
        eL, eR = extractors or (L.shim_st_ext_create(nfeatures, scale_factor, nlevels, ini_th, min_th),
                                L.shim_st_ext_create(nfeatures, scale_factor, nlevels, ini_th, min_th))
        fn, hL, hR = L.shim_st_stereo_frame, eL, eR
    else:
        L = lib()
        _declare_stereo_frame(L.ref_stereo_frame)
        made = False
        eL, eR = extractors or (RefExtractor(nfeatures, scale_factor, nlevels, ini_th, min_th),
                                RefExtractor(nfeatures, scale_factor, nlevels, ini_th, min_th))
        fn, hL, hR = L.ref_stereo_frame, eL.h, eR.h
    n = fn(hL, hR, _p(left), _p(right), w, h, left.strides[0], fx, fy, cx, cy, bf, th_depth, mb_before, _p(keys), _p(keys_un),
           _p(desc), _p(ur), _p(dep), _p(keys_r), _p(desc_r), cap, C.byref(nr), _p(cell_off), _p(cell_idx), _p(scal))
    if shim and made:
        L.shim_st_ext_destroy(eL)
        L.shim_st_ext_destroy(eR)
    if n < 0:
        raise RuntimeError(f"stereo_frame rc={n}")
    return dict(keys=keys[:n].copy(), keys_un=keys_un[:n].copy(), desc=desc[:n].copy(), u_right=ur[:n].copy(), depth=dep[:n].copy(),
                keys_right=keys_r[:nr.value].copy(), desc_right=desc_r[:nr.value].copy(), cell_off=cell_off, cell_idx=cell_idx[:n].copy(),
                scal=scal)


def _declare_frame_ctor(fn):
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    fn.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci] + [cf] * 6 + [vp] * 5 + [ci, vp, vp, vp]
    fn.restype = ci


FRAME_RGBD, FRAME_MONO, FRAME_MASKED = 0, 1, 2


def frame_ctor(kind, gray, depth_img=None, mask=None, fx=535.4, fy=539.2, cx=320.1, cy=247.6, bf=40.0, th_depth=40.0, nfeatures=1000,
               scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, shim=False, extractor=None, blur_rounding=None):
    """The reference's constructors of the TUM path, compiled (sliced verbatim):
      FRAME_RGBD    Frame(imGray, imDepth, ts, extractor, voc, K, distCoef, bf, thDepth)          src/Frame.cc:176-245
      FRAME_MONO    Frame(imGray, ts, extractor, voc, K, distCoef, bf, thDepth)                   src/Frame.cc:247-311
      FRAME_MASKED  Frame(imGray, imDepth, imMask, ts, extractor, voc, K, distCoef, bf, thDepth)  perfect/src/Frame.cc:328-420
    around the reference's extractor (shim=False, libref_orb.so) or the PRODUCT's extractor shim (shim=True, libshim_stereo.so;
    needs a GPU).  depth_img: float32 [h, w] (metres, as Tracking hands it over), mask: uint8 [h, w] of 0 / 1."""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    dimg = None if depth_img is None else np.ascontiguousarray(depth_img, np.float32)
    mimg = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    cap = nfeatures + 4 * nlevels + 256
    keys, keys_un = np.zeros(cap, KP_DTYPE), np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    ur, dep = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
    cell_off, cell_idx = np.zeros(64 * 48 + 1, np.uint32), np.zeros(cap, np.uint32)
    scal = np.zeros(8, np.float32)
    made = False
    if shim:
        L = shimstereo_lib()
        _declare_frame_ctor(L.shim_st_frame_ctor)
        fn = L.shim_st_frame_ctor
        if extractor is None:
            made = True
            extractor = L.shim_st_ext_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
            if blur_rounding is not None:
                L.shim_st_ext_set_blur_rounding(extractor, int(blur_rounding))
        hE = extractor
    else:
        L = lib()
        _declare_frame_ctor(L.ref_frame_ctor)
        fn = L.ref_frame_ctor
        extractor = extractor or RefExtractor(nfeatures, scale_factor, nlevels, ini_th, min_th)
        hE = extractor.h
    n = fn(hE, int(kind), _p(gray), None if dimg is None else _p(dimg), None if mimg is None else _p(mimg), w, h, gray.strides[0], fx, fy,
           cx, cy, bf, th_depth, _p(keys), _p(keys_un), _p(desc), _p(ur), _p(dep), cap, _p(cell_off), _p(cell_idx), _p(scal))
    if made:
        L.shim_st_ext_destroy(extractor)
    if n < 0:
        raise RuntimeError(f"frame_ctor rc={n}")
    return dict(keys=keys[:n].copy(), keys_un=keys_un[:n].copy(), desc=desc[:n].copy(), u_right=ur[:n].copy(), depth=dep[:n].copy(),
                cell_off=cell_off, cell_idx=cell_idx[:int(cell_off[-1])].copy(), scal=scal, N=n)


def configure(bump=True, canonical_trig=True, blur_mode=0):
    """The two machine-dependent spots of the reference binary and the blur column-rounding variant:
    bump=True           operator new from a bump arena -> the :686 pointer sort breaks ties by creation order
    bump=False          glibc malloc -> tie order depends on the heap's history (what a real build does)
    canonical_trig=True cos/sin at :97 = the canonical orc_sincos sequence; False = this machine's glibc cosf/sinf
    blur_mode           0 integer half-up, 1 emulate the SSE2 column kernel (SURVEY 9.4 ambiguity A)"""
    L = lib()
    L.ref_config_bump(int(bump))
    L.ref_set_trig_mode(int(canonical_trig))
    L.ref_set_blur_mode(int(blur_mode))


class RefExtractor:
    """The reference's ORB_SLAM2::ORBextractor itself (src/ORBextractor.cc), behind flat buffers."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.ref_ext_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_ext_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl, um, pat = np.zeros(n, np.int32), np.zeros(16, np.int32), np.zeros(1024, np.int32)
        self.L.ref_ext_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(fpl), _p(um), _p(pat))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=um, pattern=pat)

    def __call__(self, image, cap=None):
        """operator(): (keypoints[KP_DTYPE], descriptors[N,32])."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape if image.ndim == 2 else (0, 0)
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(-1)
        rc = self.L.ref_ext_extract(self.h, _p(image), w, h, image.strides[0] if image.ndim == 2 else 0, _p(kps),
                                    _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"ref_ext_extract rc={rc} n={n.value}")
        if n.value < 0:  # empty image: the reference returned without touching its outputs
            return None, None
        return kps[:n.value].copy(), desc[:n.value].copy()

    def keypoints_octtree(self, image, cap=None):
        """ComputePyramid + ComputeKeyPointsOctTree: list of per-level keypoint arrays (level coordinates)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        kps = np.zeros(cap, KP_DTYPE)
        nl = np.zeros(self.nlevels, np.int32)
        rc = self.L.ref_ext_keypoints_octtree(self.h, _p(image), w, h, image.strides[0], _p(kps), cap, _p(nl))
        if rc != 0:
            raise RuntimeError(f"ref_ext_keypoints_octtree rc={rc}")
        o = np.concatenate([[0], np.cumsum(nl)])
        return [kps[o[i]:o[i + 1]].copy() for i in range(self.nlevels)]

    def level(self, level, with_border=False, cap=1 << 24):
        buf = np.zeros(cap, np.uint8)
        w, h = C.c_int(), C.c_int()
        rc = self.L.ref_ext_level(self.h, level, int(with_border), _p(buf), cap, C.byref(w), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"ref_ext_level rc={rc}")
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    def candidates(self, level):
        n = self.L.ref_ext_num_candidates(self.h, level)
        out = np.zeros(n, CAND_DTYPE)
        if n:
            self.L.ref_ext_candidates(self.h, level, _p(out))
        return out

    def blurred(self):
        """GaussianBlur results of the last operator() in call order (levels with >= 1 keypoint)."""
        out = []
        for i in range(self.L.ref_ext_num_blur(self.h)):
            buf = np.zeros(1 << 24, np.uint8)
            w, h = C.c_int(), C.c_int()
            assert self.L.ref_ext_blur(self.h, i, _p(buf), buf.size, C.byref(w), C.byref(h)) == 0
            out.append(buf[:w.value * h.value].reshape(h.value, w.value).copy())
        return out

    def fast_calls(self):
        return self.L.ref_ext_fast_calls(self.h)

    def distribute_octtree(self, cands, minx, maxx, miny, maxy, N, level=0):
        cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
        out = np.zeros(max(1, cands.size), CAND_DTYPE)
        n = self.L.ref_distribute_octtree(self.h, _p(cands), cands.size, minx, maxx, miny, maxy, N, level, _p(out),
                                          out.size)
        if n < 0:
            raise RuntimeError(f"ref_distribute_octtree rc={n}")
        return out[:n].copy()


def glibc_sincosf(x):
    s, c = C.c_float(), C.c_float()
    lib().ref_glibc_sincosf(float(x), C.byref(s), C.byref(c))
    return np.float32(c.value), np.float32(s.value)


def descriptor_distance(a, b, shim=False):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return (_pick(shim, False).shim_descriptor_distance if shim else lib().ref_descriptor_distance)(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    v = [C.c_int() for _ in range(3)]
    lib().ref_three_maxima(_p(counts), counts.size, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def matcher_constants():
    v = [C.c_int() for _ in range(3)]
    lib().ref_matcher_constants(*[C.byref(x) for x in v])
    return dict(TH_LOW=v[0].value, TH_HIGH=v[1].value, HISTO_LENGTH=v[2].value)


def _csr(fv):
    node, off, idx = fv
    return (np.ascontiguousarray(node, np.uint32), np.ascontiguousarray(off, np.uint32),
            np.ascontiguousarray(idx, np.uint32))


def search_by_bow_kf_f(descKF, validKF, angKF, fvKF, descF, angF, fvF, nnratio, check_ori, shim=False):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...): (matchF2KF[nF], return value).  shim=True: through the product's
    link-level shim (HIP) instead of the reference's own body."""
    descKF = np.ascontiguousarray(descKF, np.uint8).reshape(-1, 32)
    descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
    validKF = np.ascontiguousarray(validKF, np.uint8)
    angKF = np.ascontiguousarray(angKF, np.float32)
    angF = np.ascontiguousarray(angF, np.float32)
    nk, ok, ik = _csr(fvKF)
    nf, of, if_ = _csr(fvF)
    out = np.full(descF.shape[0], -1, np.int32)
    fn = _pick(shim, False).shim_search_by_bow_kf_f if shim else lib().ref_search_by_bow_kf_f
    n = fn(_p(descKF), descKF.shape[0], _p(validKF), _p(angKF), _p(nk), _p(ok), _p(ik),
                                     nk.size, _p(descF), descF.shape[0], _p(angF), _p(nf), _p(of), _p(if_), nf.size,
                                     float(nnratio), int(check_ori), _p(out))
    return out, n


def search_by_bow_kf_kf(desc1, valid1, ang1, fv1, desc2, valid2, ang2, fv2, nnratio, check_ori, shim=False):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...): (match12[n1], return value)."""
    desc1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    desc2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    valid1 = np.ascontiguousarray(valid1, np.uint8)
    valid2 = np.ascontiguousarray(valid2, np.uint8)
    ang1 = np.ascontiguousarray(ang1, np.float32)
    ang2 = np.ascontiguousarray(ang2, np.float32)
    n1_, o1, i1 = _csr(fv1)
    n2_, o2, i2 = _csr(fv2)
    out = np.full(desc1.shape[0], -1, np.int32)
    fn = _pick(shim, False).shim_search_by_bow_kf_kf if shim else lib().ref_search_by_bow_kf_kf
    n = fn(_p(desc1), desc1.shape[0], _p(valid1), _p(ang1), _p(n1_), _p(o1), _p(i1),
                                      n1_.size, _p(desc2), desc2.shape[0], _p(valid2), _p(ang2), _p(n2_), _p(o2),
                                      _p(i2), n2_.size, float(nnratio), int(check_ori), _p(out))
    return out, n


# ---- reference bodies sliced out of Frame.cc / KeyFrame.cc / MapPoint.cc (oracle/refbuild/ref_slices.cpp) ----
def assign_grid(xy, minx, miny, gw_inv, gh_inv):
    """Frame::AssignFeaturesToGrid + PosInGrid -> (cell_off[64*48+1], cell_idx[n_in_grid])"""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    off = np.zeros(64 * 48 + 1, np.uint32)
    idx = np.zeros(max(len(xy), 1), np.uint32)
    n = lib().ref_assign_grid(_p(xy), len(xy), float(minx), float(miny), float(gw_inv), float(gh_inv), _p(off), _p(idx))
    return off, idx[:n].copy()


def features_in_area(xy, octave, off, idx, minx, miny, gw_inv, gh_inv, x, y, r, min_level=-1, max_level=-1):
    """Frame::GetFeaturesInArea on a frame whose grid is the given CSR"""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    octave = np.ascontiguousarray(octave, np.int32)
    off = np.ascontiguousarray(off, np.uint32)
    idx = np.ascontiguousarray(idx, np.uint32)
    out = np.zeros(max(len(xy), 1), np.uint32)
    n = lib().ref_features_in_area(_p(xy), _p(octave), len(xy), _p(off), _p(idx), float(minx), float(miny), float(gw_inv),
                                   float(gh_inv), float(x), float(y), float(r), int(min_level), int(max_level), _p(out), out.size)
    assert n >= 0
    return out[:n].copy()


def distinctive(pool, off, idx):
    """MapPoint::ComputeDistinctiveDescriptors per map point -> (descriptor[npoints, 32], has[npoints])"""
    pool = np.ascontiguousarray(pool, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.uint32)
    idx = np.ascontiguousarray(idx, np.uint32)
    npnt = len(off) - 1
    best = np.zeros((max(npnt, 1), 32), np.uint8)
    has = np.zeros(max(npnt, 1), np.uint8)
    lib().ref_distinctive(_p(pool), len(pool), _p(off), _p(idx), npnt, _p(best), _p(has))
    return best[:npnt], has[:npnt]


def predict_scale(max_distance, current_dist, log_scale_factor, nlevels):
    return lib().ref_predict_scale(float(max_distance), float(current_dist), float(log_scale_factor), int(nlevels))


def stereo_matches(exL, exR, kpsL, descL, kpsR, descR, mbf, mb):
    """Frame::ComputeStereoMatches (src/Frame.cc:642-846, sliced) on the pyramids of two RefExtractors' last calls"""
    kl = np.ascontiguousarray(kpsL, KP_DTYPE)
    kr = np.ascontiguousarray(kpsR, KP_DTYPE)
    dl = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    u = np.zeros(max(len(kl), 1), np.float32)
    d = np.zeros(max(len(kl), 1), np.float32)
    lib().ref_stereo_matches(exL.h, exR.h, _p(kl), _p(dl), len(kl), _p(kr), _p(dr), len(kr), float(mbf), float(mb), _p(u), _p(d))
    return u[:len(kl)], d[:len(kl)]


# ---- M4 / M9: the per-frame projection searches on mock Frames ----------------------------------------------------------
def _frame_args(cur, keep):
    """cur: dict(desc, xy, octave, angle, uRight, state, Tcw, K=(fx, fy, cx, cy, mbf, mb), bounds=(minx, maxx, miny, maxy),
    gw_inv, gh_inv, scale_factors)"""
    a = RefFrameArgs()
    arrs = dict(desc=np.ascontiguousarray(cur["desc"], np.uint8).reshape(-1, 32), xy=np.ascontiguousarray(cur["xy"], np.float32).reshape(-1, 2),
                octave=np.ascontiguousarray(cur["octave"], np.int32), angle=np.ascontiguousarray(cur["angle"], np.float32),
                uRight=np.ascontiguousarray(cur["uRight"], np.float32), state=np.ascontiguousarray(cur["state"], np.uint8),
                Tcw=np.ascontiguousarray(cur["Tcw"], np.float32).reshape(16),
                scale_factors=np.ascontiguousarray(cur["scale_factors"], np.float32))
    keep.append(arrs)
    for k, v in arrs.items():
        setattr(a, k, v.ctypes.data)
    a.n = len(arrs["desc"])
    a.fx, a.fy, a.cx, a.cy, a.mbf, a.mb = [float(v) for v in cur["K"]]
    a.minx, a.maxx, a.miny, a.maxy = [float(v) for v in cur["bounds"]]
    a.gw_inv, a.gh_inv = float(cur["gw_inv"]), float(cur["gh_inv"])
    a.nlevels = len(arrs["scale_factors"])
    return a


def search_by_projection_last_frame(cur, last, th, mono, nnratio=0.9, check_ori=True, shim=False, perfect=False, points=False):
    """ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher.cc:1578-1724).
    last: dict(Tcw, has_mp, outlier, world_pos, mpdesc, obs_gt0, octave, angle, xy).  Returns (assigned[nC], return value)
    and, with points=True (perfect/ overload :1727-1911 only), the two 2-D point lists."""
    keep = []
    a = _frame_args(cur, keep)
    n = len(last["has_mp"])
    L = dict(Tcw=np.ascontiguousarray(last["Tcw"], np.float32).reshape(16), has_mp=np.ascontiguousarray(last["has_mp"], np.uint8),
             outlier=np.ascontiguousarray(last["outlier"], np.uint8), world_pos=np.ascontiguousarray(last["world_pos"], np.float32).reshape(-1, 3),
             mpdesc=np.ascontiguousarray(last["mpdesc"], np.uint8).reshape(-1, 32), obs_gt0=np.ascontiguousarray(last["obs_gt0"], np.uint8),
             octave=np.ascontiguousarray(last["octave"], np.int32), angle=np.ascontiguousarray(last["angle"], np.float32),
             xy=np.ascontiguousarray(last["xy"], np.float32).reshape(-1, 2))
    assigned = np.full(a.n, -9, np.int32)
    pl, pc, npts = np.zeros((max(n, 1), 2), np.float32), np.zeros((max(n, 1), 2), np.float32), C.c_int32(-1)
    lb = _pick(shim, perfect)
    fn = getattr(lb, ("shim_" if shim else "ref_") + "search_by_projection_last_frame")
    rv = fn(C.byref(a), _p(L["Tcw"]), n, _p(L["has_mp"]), _p(L["outlier"]), _p(L["world_pos"]), _p(L["mpdesc"]), _p(L["obs_gt0"]),
            _p(L["octave"]), _p(L["angle"]), _p(L["xy"]), float(th), int(mono), float(nnratio), int(check_ori), _p(assigned),
            _p(pl) if points else None, _p(pc) if points else None, C.byref(npts))
    if points:
        assert npts.value >= 0, "this library has no SearchByProjection overload that returns point pairs"
        return assigned, rv, pl[:npts.value].copy(), pc[:npts.value].copy()
    return assigned, rv


def search_by_projection_local_map(cur, mps, th, nnratio=0.8, shim=False, perfect=False):
    """ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:63-157).
    mps: dict(in_view, bad, scale_level, view_cos, proj_xyr, mpdesc, obs_gt0).  Returns (assigned[nF], return value)."""
    keep = []
    a = _frame_args(cur, keep)
    n = len(mps["in_view"])
    M = dict(in_view=np.ascontiguousarray(mps["in_view"], np.uint8), bad=np.ascontiguousarray(mps["bad"], np.uint8),
             scale_level=np.ascontiguousarray(mps["scale_level"], np.int32), view_cos=np.ascontiguousarray(mps["view_cos"], np.float32),
             proj_xyr=np.ascontiguousarray(mps["proj_xyr"], np.float32).reshape(-1, 3),
             mpdesc=np.ascontiguousarray(mps["mpdesc"], np.uint8).reshape(-1, 32), obs_gt0=np.ascontiguousarray(mps["obs_gt0"], np.uint8))
    assigned = np.full(a.n, -9, np.int32)
    lb = _pick(shim, perfect)
    fn = getattr(lb, ("shim_" if shim else "ref_") + "search_by_projection_local_map")
    rv = fn(C.byref(a), n, _p(M["in_view"]), _p(M["bad"]), _p(M["scale_level"]), _p(M["view_cos"]), _p(M["proj_xyr"]), _p(M["mpdesc"]),
            _p(M["obs_gt0"]), float(th), float(nnratio), _p(assigned))
    return assigned, rv


def last_call_ms(shim=False, perfect=False):
    """wall time of the last SearchByProjection member call alone (mock construction excluded)"""
    lb = _pick(shim, perfect)
    return float(getattr(lb, ("shim_" if shim else "ref_") + "last_call_ms")())


def fuse(kf, mps, th, shim=False, perfect=False):
    """ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:1031-1182) on a mock KeyFrame.
    kf: dict(desc, xy, octave, uRight, state, obs, Rcw, tcw, Ow, K=(fx, fy, cx, cy, mbf), bounds, gw_inv, gh_inv, scale_factors,
    inv_sigma2, log_scale); mps: dict(null, bad, in_kf, world_pos, normal, max_dist, min_dist, mpdesc, obs).
    Returns (kf_assigned[nKF], mp_replaced[nmp], own_replaced[nKF], return value)."""
    f32, i32, u8 = np.float32, np.int32, np.uint8
    K = dict(desc=np.ascontiguousarray(kf["desc"], u8).reshape(-1, 32), xy=np.ascontiguousarray(kf["xy"], f32).reshape(-1, 2),
             octave=np.ascontiguousarray(kf["octave"], i32), uRight=np.ascontiguousarray(kf["uRight"], f32),
             state=np.ascontiguousarray(kf["state"], u8), obs=np.ascontiguousarray(kf["obs"], i32),
             Rcw=np.ascontiguousarray(kf["Rcw"], f32).reshape(9), tcw=np.ascontiguousarray(kf["tcw"], f32).reshape(3),
             Ow=np.ascontiguousarray(kf["Ow"], f32).reshape(3), sf=np.ascontiguousarray(kf["scale_factors"], f32),
             is2=np.ascontiguousarray(kf["inv_sigma2"], f32))
    M = dict(null=np.ascontiguousarray(mps["null"], u8), bad=np.ascontiguousarray(mps["bad"], u8), in_kf=np.ascontiguousarray(mps["in_kf"], u8),
             wp=np.ascontiguousarray(mps["world_pos"], f32).reshape(-1, 3), nr=np.ascontiguousarray(mps["normal"], f32).reshape(-1, 3),
             mx=np.ascontiguousarray(mps["max_dist"], f32), mn=np.ascontiguousarray(mps["min_dist"], f32),
             d=np.ascontiguousarray(mps["mpdesc"], u8).reshape(-1, 32), obs=np.ascontiguousarray(mps["obs"], i32))
    nKF, nmp = len(K["desc"]), len(M["null"])
    ka, mr, orr = np.full(max(nKF, 1), -9, i32), np.full(max(nmp, 1), -9, i32), np.full(max(nKF, 1), -9, i32)
    lb = _pick(shim, perfect)
    fn = getattr(lb, ("shim_" if shim else "ref_") + "fuse")
    fx, fy, cx, cy, mbf = [float(v) for v in kf["K"][:5]]
    minx, maxx, miny, maxy = [float(v) for v in kf["bounds"]]
    rv = fn(_p(K["desc"]), _p(K["xy"]), _p(K["octave"]), _p(K["uRight"]), _p(K["state"]), _p(K["obs"]), nKF, _p(K["Rcw"]), _p(K["tcw"]),
            _p(K["Ow"]), fx, fy, cx, cy, mbf, minx, maxx, miny, maxy, float(kf["gw_inv"]), float(kf["gh_inv"]), _p(K["sf"]), _p(K["is2"]),
            len(K["sf"]), float(kf["log_scale"]), nmp, _p(M["null"]), _p(M["bad"]), _p(M["in_kf"]), _p(M["wp"]), _p(M["nr"]), _p(M["mx"]),
            _p(M["mn"]), _p(M["d"]), _p(M["obs"]), float(th), _p(ka), _p(mr), _p(orr))
    return ka[:nKF], mr[:nmp], orr[:nKF], rv


class RefKfArgs(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("xy", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p), ("uRight", C.c_void_p),
                ("n", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float),
                ("minx", C.c_float), ("maxx", C.c_float), ("miny", C.c_float), ("maxy", C.c_float), ("gw_inv", C.c_float),
                ("gh_inv", C.c_float), ("scale_factors", C.c_void_p), ("inv_sigma2", C.c_void_p), ("nlevels", C.c_int),
                ("log_scale", C.c_float)]


def _pick(shim, perfect):
    if shim == "full":
        return shim_full_lib()
    return (shim_perfect_lib() if perfect else shim_lib()) if shim else (perfect_lib() if perfect else lib())


def search_by_projection_frame_kf(cur, kfp, th, orbdist, check_ori=True, shim=False, perfect=False):
    """ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, th, ORBdist) (src/ORBmatcher.cc:1757-1867,
    Tracking::Relocalization).  kfp: dict(angle, has, bad, found, world_pos, max_dist, min_dist, mpdesc).  Returns
    (assigned[nC], return value)."""
    keep = []
    a = _frame_args(cur, keep)
    f32, u8 = np.float32, np.uint8
    K = dict(angle=np.ascontiguousarray(kfp["angle"], f32), has=np.ascontiguousarray(kfp["has"], u8), bad=np.ascontiguousarray(kfp["bad"], u8),
             found=np.ascontiguousarray(kfp["found"], u8), wp=np.ascontiguousarray(kfp["world_pos"], f32).reshape(-1, 3),
             mx=np.ascontiguousarray(kfp["max_dist"], f32), mn=np.ascontiguousarray(kfp["min_dist"], f32),
             d=np.ascontiguousarray(kfp["mpdesc"], u8).reshape(-1, 32))
    assigned = np.full(max(a.n, 1), -9, np.int32)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "search_by_projection_frame_kf")
    rv = fn(C.byref(a), float(np.log(np.float32(1.2))), len(K["has"]), _p(K["angle"]), _p(K["has"]), _p(K["bad"]), _p(K["found"]), _p(K["wp"]),
            _p(K["mx"]), _p(K["mn"]), _p(K["d"]), float(th), int(orbdist), int(check_ori), _p(assigned))
    return assigned[:a.n], rv


def search_by_projection_kf_sim3(kf, Scw, pts, matched_in, th, shim=False, perfect=False):
    """ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>&, th)
    (src/ORBmatcher.cc:378-470, LoopClosing).  kf: as for fuse(); pts: dict(bad, world_pos, normal, max_dist, min_dist, mpdesc).
    Returns (matched_out[nKF], return value)."""
    f32, i32, u8 = np.float32, np.int32, np.uint8
    K = dict(desc=np.ascontiguousarray(kf["desc"], u8).reshape(-1, 32), xy=np.ascontiguousarray(kf["xy"], f32).reshape(-1, 2),
             octave=np.ascontiguousarray(kf["octave"], i32), uRight=np.ascontiguousarray(kf["uRight"], f32),
             sf=np.ascontiguousarray(kf["scale_factors"], f32), is2=np.ascontiguousarray(kf["inv_sigma2"], f32))
    a = RefKfArgs()
    a.desc, a.xy, a.octave, a.angle, a.uRight = K["desc"].ctypes.data, K["xy"].ctypes.data, K["octave"].ctypes.data, None, K["uRight"].ctypes.data
    a.n = len(K["desc"])
    a.fx, a.fy, a.cx, a.cy, a.mbf = [float(v) for v in kf["K"][:5]]
    a.minx, a.maxx, a.miny, a.maxy = [float(v) for v in kf["bounds"]]
    a.gw_inv, a.gh_inv = float(kf["gw_inv"]), float(kf["gh_inv"])
    a.scale_factors, a.inv_sigma2, a.nlevels, a.log_scale = K["sf"].ctypes.data, K["is2"].ctypes.data, len(K["sf"]), float(kf["log_scale"])
    P = dict(bad=np.ascontiguousarray(pts["bad"], u8), wp=np.ascontiguousarray(pts["world_pos"], f32).reshape(-1, 3),
             nr=np.ascontiguousarray(pts["normal"], f32).reshape(-1, 3), mx=np.ascontiguousarray(pts["max_dist"], f32),
             mn=np.ascontiguousarray(pts["min_dist"], f32), d=np.ascontiguousarray(pts["mpdesc"], u8).reshape(-1, 32))
    S = np.ascontiguousarray(Scw, f32).reshape(16)
    mi = np.ascontiguousarray(matched_in, i32)
    mo = np.full(max(a.n, 1), -9, i32)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "search_by_projection_kf_sim3")
    rv = fn(C.byref(a), _p(S), len(P["bad"]), _p(P["bad"]), _p(P["wp"]), _p(P["nr"]), _p(P["mx"]), _p(P["mn"]), _p(P["d"]), _p(mi), int(th), _p(mo))
    return mo[:a.n], rv


def _kf_args(kf, keep):
    f32, i32, u8 = np.float32, np.int32, np.uint8
    K = dict(desc=np.ascontiguousarray(kf["desc"], u8).reshape(-1, 32), xy=np.ascontiguousarray(kf["xy"], f32).reshape(-1, 2),
             octave=np.ascontiguousarray(kf["octave"], i32), angle=np.ascontiguousarray(kf["angle"], f32),
             uRight=np.ascontiguousarray(kf["uRight"], f32), sf=np.ascontiguousarray(kf["scale_factors"], f32),
             is2=np.ascontiguousarray(kf["inv_sigma2"], f32))
    keep.append(K)
    a = RefKfArgs()
    a.desc, a.xy, a.octave, a.angle, a.uRight = (K[k].ctypes.data for k in ("desc", "xy", "octave", "angle", "uRight"))
    a.n = len(K["desc"])
    a.fx, a.fy, a.cx, a.cy, a.mbf = [float(v) for v in kf["K"][:5]]
    a.minx, a.maxx, a.miny, a.maxy = [float(v) for v in kf["bounds"]]
    a.gw_inv, a.gh_inv = float(kf["gw_inv"]), float(kf["gh_inv"])
    a.scale_factors, a.inv_sigma2, a.nlevels, a.log_scale = K["sf"].ctypes.data, K["is2"].ctypes.data, len(K["sf"]), float(kf["log_scale"])
    return a


def search_for_triangulation(k1, k2, F12, only_stereo, check_ori=True, shim=False, perfect=False):
    """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:827-1012) on two mock KeyFrames.  k1 / k2: keyframe dicts as
    fuse() takes them + angle, has_mp, fv = (node, off, idx); k1["Ow"], k2["Rcw"], k2["tcw"], k2["level_sigma2"].
    Returns (pairs[n, 2], return value)."""
    keep = []
    a, b = _kf_args(k1, keep), _kf_args(k2, keep)
    u8, u32, f32 = np.uint8, np.uint32, np.float32
    h1, h2 = np.ascontiguousarray(k1["has_mp"], u8), np.ascontiguousarray(k2["has_mp"], u8)
    fv1 = [np.ascontiguousarray(x, u32) for x in k1["fv"]]
    fv2 = [np.ascontiguousarray(x, u32) for x in k2["fv"]]
    Ow, Rcw, tcw = (np.ascontiguousarray(x, f32).reshape(-1) for x in (k1["Ow"], k2["Rcw"], k2["tcw"]))
    ls2 = np.ascontiguousarray(k2["level_sigma2"], f32)
    F = np.ascontiguousarray(F12, f32).reshape(9)
    cap = max(a.n, 1)
    pairs = np.full((cap, 2), -1, np.int32)
    npairs = C.c_int32(0)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "search_for_triangulation")
    rv = fn(C.byref(a), _p(h1), _p(fv1[0]), _p(fv1[1]), _p(fv1[2]), len(fv1[0]), _p(Ow), C.byref(b), _p(h2), _p(fv2[0]), _p(fv2[1]), _p(fv2[2]),
            len(fv2[0]), _p(Rcw), _p(tcw), _p(ls2), _p(F), int(only_stereo), int(check_ori), _p(pairs), cap, C.byref(npairs))
    return pairs[:npairs.value].copy(), rv


def fuse_sim3(kf, Scw, mps, th, shim=False, perfect=False):
    """ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, th, vector<MapPoint*> &vpReplacePoint)
    (src/ORBmatcher.cc:1198-1299).  kf / mps as fuse().  Returns (kf_assigned[nKF], replace_point[np], return value)."""
    keep = []
    kk = dict(kf)
    kk.setdefault("angle", np.zeros(len(np.asarray(kf["octave"])), np.float32))
    a = _kf_args(kk, keep)
    f32, u8 = np.float32, np.uint8
    st = np.ascontiguousarray(kf["state"], u8)
    M = dict(null=np.ascontiguousarray(mps["null"], u8), bad=np.ascontiguousarray(mps["bad"], u8),
             wp=np.ascontiguousarray(mps["world_pos"], f32).reshape(-1, 3), nr=np.ascontiguousarray(mps["normal"], f32).reshape(-1, 3),
             mx=np.ascontiguousarray(mps["max_dist"], f32), mn=np.ascontiguousarray(mps["min_dist"], f32),
             d=np.ascontiguousarray(mps["mpdesc"], u8).reshape(-1, 32))
    S = np.ascontiguousarray(Scw, f32).reshape(16)
    n = len(M["null"])
    ka, rp = np.full(max(a.n, 1), -9, np.int32), np.full(max(n, 1), -9, np.int32)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "fuse_sim3")
    rv = fn(C.byref(a), _p(st), _p(S), n, _p(M["null"]), _p(M["bad"]), _p(M["wp"]), _p(M["nr"]), _p(M["mx"]), _p(M["mn"]), _p(M["d"]), float(th),
            _p(ka), _p(rp))
    return ka[:a.n], rp[:n], rv


class RefKfPoints(C.Structure):
    _fields_ = [("state", C.c_void_p), ("world_pos", C.c_void_p), ("max_dist", C.c_void_p), ("min_dist", C.c_void_p),
                ("mpdesc", C.c_void_p), ("Rcw", C.c_void_p), ("tcw", C.c_void_p)]


def _kf_points(kf, keep):
    f32, u8 = np.float32, np.uint8
    K = dict(state=np.ascontiguousarray(kf["state"], u8), wp=np.ascontiguousarray(kf["world_pos"], f32).reshape(-1, 3),
             mx=np.ascontiguousarray(kf["max_dist"], f32), mn=np.ascontiguousarray(kf["min_dist"], f32),
             d=np.ascontiguousarray(kf["mpdesc"], u8).reshape(-1, 32), R=np.ascontiguousarray(kf["Rcw"], f32).reshape(9),
             t=np.ascontiguousarray(kf["tcw"], f32).reshape(3))
    keep.append(K)
    p = RefKfPoints()
    p.state, p.world_pos, p.max_dist, p.min_dist, p.mpdesc, p.Rcw, p.tcw = (K[k].ctypes.data for k in ("state", "wp", "mx", "mn", "d", "R", "t"))
    return p


def search_by_sim3(k1, k2, s12, R12, t12, th, matches_in, shim=False, perfect=False):
    """ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:1334-1548) on two mock KeyFrames.  k1 / k2: keyframe dicts as fuse() takes
    them + per-feature point arrays state / world_pos / max_dist / min_dist / mpdesc and the pose Rcw, tcw.  matches_in[N1]:
    -1 NULL, j >= 0 the point of k2's feature j, -2 a point k2 does not observe.  Returns (matches_out[N1], return value)."""
    keep = []
    ka, kb = dict(k1), dict(k2)
    for k in (ka, kb):
        k.setdefault("angle", np.zeros(len(np.asarray(k["octave"])), np.float32))
    a, b = _kf_args(ka, keep), _kf_args(kb, keep)
    pa, pb = _kf_points(k1, keep), _kf_points(k2, keep)
    R = np.ascontiguousarray(R12, np.float32).reshape(9)
    t = np.ascontiguousarray(t12, np.float32).reshape(3)
    m = np.ascontiguousarray(matches_in, np.int32).copy()
    if len(m) == 0:
        m = np.zeros(1, np.int32)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "search_by_sim3")
    rv = fn(C.byref(a), C.byref(pa), C.byref(b), C.byref(pb), float(s12), _p(R), _p(t), float(th), _p(m))
    return m[:a.n], rv


def search_for_initialization(f1, f2, prev_matched, window, nnratio=0.9, check_ori=True, shim=False, perfect=False):
    """ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:523-651) on two mock Frames (dicts as _frame_args takes them).
    Returns (matches12[n1], prev_matched_out[n1, 2], return value)."""
    keep = []
    a, b = _frame_args(f1, keep), _frame_args(f2, keep)
    pm = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
    if len(pm) == 0:
        pm = np.zeros((1, 2), np.float32)
    m = np.full(max(a.n, 1), -9, np.int32)
    fn = getattr(_pick(shim, perfect), ("shim_" if shim else "ref_") + "search_for_initialization")
    rv = fn(C.byref(a), C.byref(b), _p(pm), int(window), float(nnratio), int(bool(check_ori)), _p(m))
    return m[:a.n], pm[:a.n], rv


# ---- the reference's binary map file: Map::Save / Map::Load (perfect/src/Map.cc:143-430, sliced; libref_perfect.so) ----
def _map_lib():
    L = perfect_lib()
    vp, ci = C.c_void_p, C.c_int
    L.ref_map_save.argtypes = [C.c_char_p, ci, vp, vp, ci] + [vp] * 12
    L.ref_map_load.restype = vp
    L.ref_map_load.argtypes = [C.c_char_p]
    L.ref_map_loaded_free.argtypes = [vp]
    L.ref_map_loaded_free.restype = None
    L.ref_map_loaded_counts.argtypes = [vp] * 7
    L.ref_map_loaded_counts.restype = None
    L.ref_map_loaded_get.argtypes = [vp] * 20
    L.ref_map_loaded_get.restype = None
    return L


def map_save(path, mappoints, keyframes):
    """Map::Save itself on a map built from flat data.  mappoints: [(id, (x, y, z))]; keyframes: dicts with id, timestamp,
    t_cw, q_cw, kps (KP_DTYPE), desc, mp_index (index into mappoints or ULONG_MAX), parent (keyframe id or None),
    connections [(keyframe id, weight)].  (The quaternion rides through the mock pose unchanged: Eigen is not vendored.)"""
    L = _map_lib()
    nmp, nkf = len(mappoints), len(keyframes)
    mp_id = np.array([m[0] for m in mappoints] or [0], np.uint64)
    mp_pos = np.array([m[1] for m in mappoints] or [(0, 0, 0)], np.float32).reshape(-1, 3)
    idx_of = {kf["id"]: i for i, kf in enumerate(keyframes)}
    kf_id = np.array([kf["id"] for kf in keyframes] or [0], np.uint64)
    kf_ts = np.array([kf["timestamp"] for kf in keyframes] or [0], np.float64)
    kf_t = np.array([kf["t_cw"] for kf in keyframes] or [(0, 0, 0)], np.float32).reshape(-1, 3)
    kf_q = np.array([kf["q_cw"] for kf in keyframes] or [(0, 0, 0, 1)], np.float32).reshape(-1, 4)
    kf_n = np.array([len(kf["kps"]) for kf in keyframes] or [0], np.int32)
    kps = np.concatenate([np.ascontiguousarray(kf["kps"], KP_DTYPE) for kf in keyframes] + [np.zeros(1, KP_DTYPE)])
    desc = np.concatenate([np.ascontiguousarray(kf["desc"], np.uint8).reshape(-1, 32) for kf in keyframes] + [np.zeros((1, 32), np.uint8)])
    mpi = np.concatenate([np.where(np.asarray(kf["mp_index"], np.uint64) == np.uint64(0xFFFFFFFFFFFFFFFF), -1,
                                   np.asarray(kf["mp_index"], np.uint64).astype(np.int64)).astype(np.int64) for kf in keyframes]
                         + [np.zeros(1, np.int64)])
    parent = np.array([-1 if kf.get("parent") is None else idx_of[kf["parent"]] for kf in keyframes] or [-1], np.int64)
    con_off = np.zeros(nkf + 1, np.int32)
    con_kf, con_w = [], []
    for i, kf in enumerate(keyframes):
        for cid, wgt in kf.get("connections", []):
            con_kf.append(idx_of[cid])
            con_w.append(wgt)
        con_off[i + 1] = len(con_kf)
    con_kf, con_w = np.array(con_kf or [0], np.int32), np.array(con_w or [0], np.int32)
    rc = L.ref_map_save(str(path).encode(), nmp, _p(mp_id), _p(mp_pos), nkf, _p(kf_id), _p(kf_ts), _p(kf_t), _p(kf_q), _p(kf_n), _p(kps),
                        _p(desc), _p(mpi), _p(parent), _p(con_off), _p(con_kf), _p(con_w))
    if rc != 0:
        raise RuntimeError("Map::Save failed")


def map_load(path, bump=True):
    """Map::Load itself; returns (mappoints, keyframes, info) in FILE order, the same shapes as mapio.load_map plus what the
    reference did on the way: info = dict(log, next_mp_id, mp_set_rank, mp_nobs, mp_calls).  bump: run it with the bump
    allocator (heap addresses grow with creation order) -- the reference resolves the stored map-point indices through a
    std::set<MapPoint*>, so under glibc malloc the links it reads back depend on the heap's history."""
    L = _map_lib()
    L.ref_config_bump(int(bump))
    h = L.ref_map_load(str(path).encode())
    if not h:
        raise RuntimeError("Map::Load failed")
    try:
        v = [C.c_int() for _ in range(5)]
        nxt = C.c_uint64()
        L.ref_map_loaded_counts(h, *[C.byref(x) for x in v], C.byref(nxt))
        nmp, nkf, nfeat, ncon, loglen = (x.value for x in v)
        mp_id, mp_pos = np.zeros(max(nmp, 1), np.uint64), np.zeros((max(nmp, 1), 3), np.float32)
        rank, nobs, calls = np.zeros(max(nmp, 1), np.int32), np.zeros(max(nmp, 1), np.int32), np.zeros((max(nmp, 1), 2), np.int32)
        kf_id, kf_ts = np.zeros(max(nkf, 1), np.uint64), np.zeros(max(nkf, 1), np.float64)
        kf_t, kf_q, kf_n = np.zeros((max(nkf, 1), 3), np.float32), np.zeros((max(nkf, 1), 4), np.float32), np.zeros(max(nkf, 1), np.int32)
        kps, desc = np.zeros(max(nfeat, 1), KP_DTYPE), np.zeros((max(nfeat, 1), 32), np.uint8)
        kf_mp, urd = np.zeros(max(nfeat, 1), np.int64), np.zeros((max(nfeat, 1), 2), np.float32)
        par, con_off = np.zeros(max(nkf, 1), np.uint64), np.zeros(nkf + 1, np.int32)
        con_id, con_w = np.zeros(max(ncon, 1), np.uint64), np.zeros(max(ncon, 1), np.int32)
        log = C.create_string_buffer(loglen + 1)
        L.ref_map_loaded_get(h, _p(mp_id), _p(mp_pos), _p(rank), _p(nobs), _p(calls), _p(kf_id), _p(kf_ts), _p(kf_t), _p(kf_q), _p(kf_n),
                             _p(kps), _p(desc), _p(kf_mp), _p(urd), _p(par), _p(con_off), _p(con_id), _p(con_w), log)
    finally:
        L.ref_map_loaded_free(h)
    mps = [(int(mp_id[i]), tuple(float(c) for c in mp_pos[i])) for i in range(nmp)]
    kfs, at = [], 0
    for k in range(nkf):
        n = int(kf_n[k])
        kfs.append(dict(id=int(kf_id[k]), timestamp=float(kf_ts[k]), t_cw=kf_t[k].copy(), q_cw=kf_q[k].copy(), kps=kps[at:at + n].copy(),
                        desc=desc[at:at + n].copy(), mp_id=kf_mp[at:at + n].copy(), u_right=urd[at:at + n, 0].copy(), depth=urd[at:at + n, 1].copy(),
                        parent=None if par[k] == np.uint64(0xFFFFFFFFFFFFFFFF) else int(par[k]),
                        connections=[(int(con_id[c]), int(con_w[c])) for c in range(con_off[k], con_off[k + 1])]))
        at += n
    info = dict(log=log.raw[:loglen].decode(), next_mp_id=int(nxt.value), mp_set_rank=rank[:nmp].copy(), mp_nobs=nobs[:nmp].copy(),
                mp_calls=calls[:nmp].copy())
    return mps, kfs, info


# ---- the DBoW2 twin (oracle/refbuild/dbow2_twin: TemplatedVocabulary / FORB / BowVector / FeatureVector in DBoW2's class shape) ----
_TWIN = os.path.join(_OUT, "libdbow2_twin.so")
TEXT2BINARY = os.path.join(_OUT, "text2binary")   # the reference's tool/text2binary.cc, compiled UNCHANGED against the twin
_twin = None


def twin_available():
    return (os.path.exists(_TWIN) and os.path.exists(TEXT2BINARY)) or have_reference()


def twin_lib():
    global _twin
    if _twin is None:
        if have_reference():
            _make("twin")
        if not os.path.exists(_TWIN):
            raise RuntimeError("oracle/_ref/libdbow2_twin.so is missing and /root/reference is not available to build it")
        L = C.CDLL(_TWIN)
        vp, ci = C.c_void_p, C.c_int
        L.twin_voc_load.restype = vp
        L.twin_voc_load.argtypes = [C.c_char_p, ci]
        L.twin_voc_free.argtypes = [vp]
        L.twin_voc_save_binary.argtypes = [vp, C.c_char_p]
        L.twin_voc_info.argtypes = [vp] * 7
        L.twin_voc_arrays.argtypes = [vp] * 6
        L.twin_voc_transform.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp]
        L.twin_voc_score.restype = C.c_double
        L.twin_voc_score.argtypes = [vp, vp, vp, ci, vp, vp, ci]
        _twin = L
    return _twin


class TwinVocabulary:
    """ORB_SLAM2::ORBVocabulary (the reference's typedef, include/ORBVocabulary.h:16-17) instantiated over the DBoW2 twin"""

    def __init__(self, path, binary=False):
        self.L = twin_lib()
        self.h = self.L.twin_voc_load(str(path).encode(), int(binary))
        if not self.h:
            raise RuntimeError(f"twin vocabulary: cannot load {path}")
        v = [C.c_int() for _ in range(6)]
        self.L.twin_voc_info(self.h, *[C.byref(x) for x in v])
        self.k, self.depth, self.nnodes, self.nwords, self.scoring, self.weighting = (x.value for x in v)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.twin_voc_free(self.h)
            self.h = None

    def save_binary(self, path):
        self.L.twin_voc_save_binary(self.h, str(path).encode())

    def arrays(self):
        nn = self.nnodes
        parent, leaf, desc = np.zeros(nn, np.uint32), np.zeros(nn, np.uint8), np.zeros((nn, 32), np.uint8)
        weight, word = np.zeros(nn, np.float64), np.zeros(nn, np.uint32)
        self.L.twin_voc_arrays(self.h, _p(parent), _p(leaf), _p(desc), _p(weight), _p(word))
        return dict(parent=parent, is_leaf=leaf, node_desc=desc, weight=weight, word_id=word)

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        bid, bval, nb = np.zeros(m, np.uint32), np.zeros(m, np.float64), C.c_int()
        fvn, fvo, fvi, nf = np.zeros(m, np.uint32), np.zeros(m + 1, np.uint32), np.zeros(m, np.uint32), C.c_int()
        self.L.twin_voc_transform(self.h, _p(desc), n, levelsup, _p(bid), _p(bval), C.byref(nb), _p(fvn), _p(fvo), _p(fvi), C.byref(nf))
        return dict(bow_id=bid[:nb.value].copy(), bow_val=bval[:nb.value].copy(), fv_node=fvn[:nf.value].copy(),
                    fv_off=fvo[:nf.value + 1].copy(), fv_idx=fvi[:int(fvo[nf.value])].copy())

    def score(self, a, b):
        ia, va = np.ascontiguousarray(a[0], np.uint32), np.ascontiguousarray(a[1], np.float64)
        ib, vb = np.ascontiguousarray(b[0], np.uint32), np.ascontiguousarray(b[1], np.float64)
        return float(self.L.twin_voc_score(self.h, _p(ia), _p(va), len(ia), _p(ib), _p(vb), len(ib)))
