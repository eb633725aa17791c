"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (orb_slam2_ssd_semantic_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liborb_oracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
CAND_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4")])
assert KP_DTYPE.itemsize == 28 and CAND_DTYPE.itemsize == 12


def build(force=False):
    """Compile the oracle with gcc (never the reference's build system)."""
    src = [os.path.join(_HERE, f) for f in ("orb_oracle.c", "orb_oracle.h", "orc_pattern.inc")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborb_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    u8p, i32p, f32p, u32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                             C.POINTER(C.c_uint32))
    vp = C.c_void_p
    L.orc_create.restype = vp
    L.orc_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
    L.orc_destroy.argtypes = [vp]
    L.orc_nlevels.argtypes = [vp]
    L.orc_get_scales.argtypes = [vp, vp, vp, vp, vp]
    L.orc_get_features_per_level.argtypes = [vp, vp]
    L.orc_get_umax.argtypes = [vp]
    L.orc_get_pattern.restype = vp
    L.orc_level_sizes.argtypes = [vp, C.c_int, C.c_int, vp, vp]
    L.orc_cell_grid.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp]
    L.orc_resize_linear_u8.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    L.orc_resize_tables.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp]
    L.orc_copy_make_border101.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int]
    L.orc_fast9.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orc_fast_score_map.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
    L.orc_distribute_octtree.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                         C.c_int, vp]
    L.orc_fast_atan2.restype = C.c_float
    L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.orc_ic_moments.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp]
    L.orc_ic_angle.restype = C.c_float
    L.orc_ic_angle.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.orc_gaussian_blur7.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp]
    L.orc_sincos.argtypes = [C.c_float, vp, vp]
    L.orc_descriptor.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
    L.orc_extract.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp]
    for f in ("orc_tap_level", "orc_tap_blurred"):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp, C.c_int, vp, vp, vp]
    for f in ("orc_tap_candidates", "orc_tap_selected"):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp, C.c_int, vp]
    L.orc_tap_blur_ties.restype = C.c_long
    L.orc_tap_blur_ties.argtypes = [vp]
    L.orc_tap_octree_tie_breaks.argtypes = [vp]
    L.orc_set_blur_mode.argtypes = [vp, C.c_int]
    L.orc_hamming.argtypes = [vp, vp]
    L.orc_three_maxima.argtypes = [vp, C.c_int, vp, vp, vp]
    L.orc_rot_bin.argtypes = [C.c_float, C.c_float]
    L.orc_match_bf.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp]
    L.orc_search_by_bow.argtypes = ([vp, C.c_int, vp, vp, vp, vp, vp, C.c_int] * 2
                                    + [C.c_float, C.c_int, C.c_int, C.c_int, vp, vp])
    L.orc_hamming_csr.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp]
    L.orc_hamming_csr2.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp]
    L.orc_assign_grid.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, vp, vp]
    L.orc_features_in_area.argtypes = [vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_int, C.c_int, vp, C.c_int]
    L.orc_stereo_matches.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
    L.orc_bow_transform.argtypes = [C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int] + [vp] * 6 + [vp] * 4
    L.orc_distinctive.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp]
    cf, ci = C.c_float, C.c_int
    L.orc_search_by_projection.argtypes = [vp, vp, vp, ci, vp, vp, cf, cf, cf, cf, vp, vp, vp, vp, ci, ci, cf, ci, vp, vp, vp]
    L.orc_search_by_projection_chi2.argtypes = [vp, vp, vp, ci, vp, vp, cf, cf, cf, cf, vp, vp, vp, ci, vp, vp, ci, ci, cf, ci, vp, vp, vp]
    L.orc_proj_queries_last_frame.argtypes = [vp, vp] + [cf] * 10 + [vp, ci, vp, vp, vp, vp, vp, cf, ci, ci, vp, vp]
    L.orc_proj_queries_local_map.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, cf, vp, vp]
    L.orc_search_for_triangulation.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, cf, cf, vp, vp, ci, vp]
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


class OracleExtractor:
    """CPU twin of ORB_SLAM2::ORBextractor (include/ORBextractor.h:35-116)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("orc_create failed")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def scales(self):
        out = [np.zeros(self.nlevels, np.float32) for _ in range(4)]
        self.L.orc_get_scales(self.h, *[_p(o) for o in out])
        return out

    def features_per_level(self):
        out = np.zeros(self.nlevels, np.int32)
        self.L.orc_get_features_per_level(self.h, _p(out))
        return out

    def level_sizes(self, w, h):
        lw = np.zeros(self.nlevels, np.int32)
        lh = np.zeros(self.nlevels, np.int32)
        self.L.orc_level_sizes(self.h, w, h, _p(lw), _p(lh))
        return lw, lh

    def set_blur_mode(self, mode):
        self.L.orc_set_blur_mode(self.h, mode)

    def __call__(self, image, cap=None):
        """operator(): returns (keypoints[KP_DTYPE], descriptors[N,32] u8)."""
        image = _u8(image)
        h, w = image.shape
        cap = cap or (self.nfeatures + 4 * self.nlevels + 64)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = self.L.orc_extract(self.h, _p(image), w, h, image.strides[0], _p(kps), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"orc_extract rc={rc} n={n.value}")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def _tap_img(self, fn, level):
        w, h, s = C.c_int(), C.c_int(), C.c_int()
        p = fn(self.h, level, C.byref(w), C.byref(h), C.byref(s))
        if not p:
            return None
        buf = (C.c_uint8 * (s.value * h.value)).from_address(p)
        return np.frombuffer(buf, np.uint8).reshape(h.value, s.value)[:, :w.value].copy()

    def level(self, level):
        return self._tap_img(self.L.orc_tap_level, level)

    def blurred(self, level):
        return self._tap_img(self.L.orc_tap_blurred, level)

    def _tap_cand(self, fn, level):
        n = C.c_int()
        p = fn(self.h, level, C.byref(n))
        if not p or n.value == 0:
            return np.zeros(0, CAND_DTYPE)
        buf = (C.c_uint8 * (12 * n.value)).from_address(p)
        return np.frombuffer(buf, CAND_DTYPE).copy()

    def candidates(self, level):
        return self._tap_cand(self.L.orc_tap_candidates, level)

    def selected(self, level):
        return self._tap_cand(self.L.orc_tap_selected, level)

    def blur_ties(self):
        return self.L.orc_tap_blur_ties(self.h)

    def octree_tie_breaks(self):
        return self.L.orc_tap_octree_tie_breaks(self.h)


# ---- stage-level free functions -------------------------------------------------------------------
def umax():
    out = np.zeros(16, np.int32)
    lib().orc_get_umax(_p(out))
    return out


def pattern():
    p = lib().orc_get_pattern()
    return np.frombuffer((C.c_int8 * 1024).from_address(p), np.int8).copy()


def cell_grid(lw, lh):
    v = [C.c_int() for _ in range(4)]
    ok = lib().orc_cell_grid(lw, lh, *[C.byref(x) for x in v])
    return (ok,) + tuple(x.value for x in v)


def resize_linear(src, dw, dh):
    src = _u8(src)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def resize_tables(ssize, dsize, is_x):
    ofs = np.zeros(dsize, np.int32)
    coef = np.zeros(2 * dsize, np.int16)
    lib().orc_resize_tables(ssize, dsize, int(is_x), _p(ofs), _p(coef))
    return ofs, coef.reshape(dsize, 2)


def copy_make_border101(src, border):
    src = _u8(src)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().orc_copy_make_border101(_p(src), w, h, src.strides[0], _p(dst), dst.strides[0], border)
    return dst


def fast9(img, threshold, nonmax=True):
    img = _u8(img)
    h, w = img.shape
    out = np.zeros(max(1, w * h), CAND_DTYPE)
    n = lib().orc_fast9(_p(img), w, h, img.strides[0], threshold, int(nonmax), _p(out), out.size)
    assert n >= 0
    return out[:n].copy()


def fast_score_map(img):
    img = _u8(img)
    h, w = img.shape
    sc = np.zeros((h, w), np.uint8)
    lib().orc_fast_score_map(_p(img), w, h, img.strides[0], _p(sc), w)
    return sc


def distribute_octtree(cands, minx, maxx, miny, maxy, N):
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    out = np.zeros(max(1, cands.size), CAND_DTYPE)
    st = np.zeros(3, np.int32)
    n = lib().orc_distribute_octtree(_p(cands), cands.size, minx, maxx, miny, maxy, N, _p(out), out.size, _p(st))
    if n < 0:
        raise RuntimeError(f"orc_distribute_octtree rc={n}")
    return out[:n].copy(), dict(iterations=int(st[0]), phaseb_passes=int(st[1]), tie_breaks=int(st[2]))


def fast_atan2(y, x):
    return np.float32(lib().orc_fast_atan2(float(y), float(x)))


def ic_moments(img, x, y):
    img = _u8(img)
    a, b = C.c_int(), C.c_int()
    lib().orc_ic_moments(_p(img), img.strides[0], x, y, C.byref(a), C.byref(b))
    return a.value, b.value


def ic_angle(img, x, y):
    img = _u8(img)
    return np.float32(lib().orc_ic_angle(_p(img), img.strides[0], x, y))


def gaussian_blur7(img, mode=0):
    img = _u8(img)
    h, w = img.shape
    dst = np.zeros((h, w), np.uint8)
    ties = C.c_long()
    lib().orc_gaussian_blur7(_p(img), w, h, img.strides[0], _p(dst), w, mode, C.byref(ties))
    return dst, ties.value


def sincos(angle_deg):
    a, b = C.c_float(), C.c_float()
    lib().orc_sincos(float(angle_deg), C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def descriptor(blurred, x, y, angle_deg):
    blurred = _u8(blurred)
    d = np.zeros(32, np.uint8)
    lib().orc_descriptor(_p(blurred), blurred.strides[0], x, y, float(angle_deg), _p(d))
    return d


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_hamming(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    i = [C.c_int() for _ in range(3)]
    lib().orc_three_maxima(_p(counts), counts.size, *[C.byref(x) for x in i])
    return tuple(x.value for x in i)


def rot_bin(a1, a2):
    return lib().orc_rot_bin(float(a1), float(a2))


def match_bf(q, t, q_angle=None, t_angle=None, nnratio=0.9, th=100, check_ori=True):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    qa = None if q_angle is None else np.ascontiguousarray(q_angle, np.float32)
    ta = None if t_angle is None else np.ascontiguousarray(t_angle, np.float32)
    m = np.full(len(q), -1, np.int32)
    b = np.zeros(len(q), np.int32)
    s = np.zeros(len(q), np.int32)
    n = C.c_int()
    rc = lib().orc_match_bf(_p(q), len(q), _p(t), len(t), _p(qa), _p(ta), nnratio, th, int(check_ori), _p(m),
                            _p(b), _p(s), C.byref(n))
    assert rc == 0
    return m, b, s, n.value


def search_by_bow(descKF, validKF, angKF, fvKF, descF, validF, angF, fvF, nnratio=0.7, th_low=50,
                  strict_lt=False, check_ori=True):
    """fvKF / fvF = (node[nn], off[nn+1], idx[...]) CSR feature vectors with ascending node ids."""
    descKF = np.ascontiguousarray(descKF, np.uint8).reshape(-1, 32)
    descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
    vk = None if validKF is None else np.ascontiguousarray(validKF, np.uint8)
    vf = None if validF is None else np.ascontiguousarray(validF, np.uint8)
    ak = np.ascontiguousarray(angKF, np.float32)
    af = np.ascontiguousarray(angF, np.float32)
    nk, ok, ik = [np.ascontiguousarray(a, np.uint32) for a in fvKF]
    nf, of, if_ = [np.ascontiguousarray(a, np.uint32) for a in fvF]
    m = np.full(len(descF), -1, np.int32)
    n = C.c_int()
    rc = lib().orc_search_by_bow(_p(descKF), len(descKF), _p(vk), _p(ak), _p(nk), _p(ok), _p(ik), len(nk),
                                 _p(descF), len(descF), _p(vf), _p(af), _p(nf), _p(of), _p(if_), len(nf),
                                 nnratio, th_low, int(strict_lt), int(check_ori), _p(m), C.byref(n))
    assert rc == 0
    return m, n.value


def hamming_csr(q, t, off, cand):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.uint32)
    cand = np.ascontiguousarray(cand, np.uint32)
    bi = np.zeros(len(q), np.int32)
    b = np.zeros(len(q), np.int32)
    s = np.zeros(len(q), np.int32)
    rc = lib().orc_hamming_csr(_p(q), len(q), _p(t), len(t), _p(off), _p(cand), _p(bi), _p(b), _p(s))
    assert rc == 0
    return bi, b, s


def hamming_csr2(q, t, off, cand):
    """hamming_csr plus the candidate that owns the runner-up distance (bestLevel2's owner at ORBmatcher.cc:128-140)"""
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.uint32)
    cand = np.ascontiguousarray(cand, np.uint32)
    bi, b, s, si = (np.zeros(len(q), np.int32) for _ in range(4))
    rc = lib().orc_hamming_csr2(_p(q), len(q), _p(t), len(t), _p(off), _p(cand), _p(bi), _p(b), _p(s), _p(si))
    assert rc == 0
    return bi, b, s, si


def assign_grid(xy, minx, miny, gw_inv, gh_inv):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    off = np.zeros(64 * 48 + 1, np.uint32)
    idx = np.zeros(max(len(xy), 1), np.uint32)
    n = lib().orc_assign_grid(_p(xy), len(xy), minx, miny, gw_inv, gh_inv, _p(off), _p(idx))
    return off, idx[:n].copy()


def features_in_area(xy, octave, off, idx, minx, miny, gw_inv, gh_inv, x, y, r, min_level=-1, max_level=-1):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    octave = np.ascontiguousarray(octave, np.int32)
    off = np.ascontiguousarray(off, np.uint32)
    idx = np.ascontiguousarray(idx, np.uint32)
    out = np.zeros(max(len(xy), 1), np.uint32)
    n = lib().orc_features_in_area(_p(xy), _p(octave), _p(off), _p(idx), minx, miny, gw_inv, gh_inv, x, y, r, min_level,
                                   max_level, _p(out), out.size)
    assert n >= 0
    return out[:n].copy()


def distinctive(pool, off, idx):
    pool = np.ascontiguousarray(pool, np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(off, np.uint32)
    idx = np.ascontiguousarray(idx, np.uint32)
    npts = len(off) - 1
    best = np.zeros(max(npts, 1), np.int32)
    med = np.zeros(max(npts, 1), np.int32)
    rc = lib().orc_distinctive(_p(pool), len(pool), _p(off), _p(idx), npts, _p(best), _p(med))
    assert rc == 0
    return best[:npts].copy(), med[:npts].copy()


def stereo_matches(exL, exR, kpsL, descL, kpsR, descR, mbf, mb):
    """Frame::ComputeStereoMatches; exL / exR are OracleExtractors whose last call saw the left / right image."""
    kpsL = np.ascontiguousarray(kpsL, KP_DTYPE)
    kpsR = np.ascontiguousarray(kpsR, KP_DTYPE)
    descL = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
    descR = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
    n = len(kpsL)
    u = np.zeros(max(n, 1), np.float32)
    d = np.zeros(max(n, 1), np.float32)
    sad = np.zeros(max(n, 1), np.int32)
    rc = lib().orc_stereo_matches(exL.h, exR.h, _p(kpsL), _p(descL), n, _p(kpsR), _p(descR), len(kpsR), mbf, mb, _p(u),
                                  _p(d), _p(sad))
    assert rc == 0
    return u[:n].copy(), d[:n].copy(), sad[:n].copy()


def bow_transform(voc, desc, levelsup=4):
    """voc: dict(child_off, child_idx, node_desc, word_id, weight, L).  Returns dict of per-feature word/node/weight,
    the BowVector (ids, values) and the FeatureVector CSR (node, off, idx)."""
    co = np.ascontiguousarray(voc["child_off"], np.uint32)
    ci = np.ascontiguousarray(voc["child_idx"], np.uint32)
    nd = np.ascontiguousarray(voc["node_desc"], np.uint8).reshape(-1, 32)
    wi = np.ascontiguousarray(voc["word_id"], np.uint32)
    ww = np.ascontiguousarray(voc["weight"], np.float64)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(desc)
    m = max(n, 1)
    fw, fn, fwt = np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros(m, np.float64)
    bid, bval, nb = np.zeros(m, np.uint32), np.zeros(m, np.float64), C.c_int(0)
    fvn, fvo, fvi, nf = np.zeros(m, np.uint32), np.zeros(m + 1, np.uint32), np.zeros(m, np.uint32), C.c_int(0)
    rc = lib().orc_bow_transform(len(nd), _p(co), _p(ci), _p(nd), _p(wi), _p(ww), int(voc["L"]), levelsup, _p(desc), n,
                                 _p(fw), _p(fn), _p(fwt), _p(bid), _p(bval), C.byref(nb), _p(fvn), _p(fvo), _p(fvi),
                                 C.byref(nf))
    assert rc == 0
    return dict(word=fw[:n].copy(), node=fn[:n].copy(), weight=fwt[:n].copy(), bow_id=bid[:nb.value].copy(),
                bow_val=bval[:nb.value].copy(), fv_node=fvn[:nf.value].copy(), fv_off=fvo[:nf.value + 1].copy(),
                fv_idx=fvi[:int(fvo[nf.value])].copy())


# ---- M4: projection-gated searches ------------------------------------------------------------------------------------
PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("ur", "<f4"), ("flags", "<i4"), ("pad", "<i4")])


def search_by_projection(descF, xyF, octF, grid, bounds, uRight, blocked, queries, qdesc, th, nnratio, ratio_rule,
                         inv_level_sigma2=None):
    """orc_search_by_projection(_chi2): grid = (cell_off, cell_idx), bounds = (minx, miny, gw_inv, gh_inv); inv_level_sigma2
    switches the reprojection-error gate on for the queries with flag 4; returns (match[nq], best[nq], second[nq])"""
    descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
    xyF = np.ascontiguousarray(xyF, np.float32).reshape(-1, 2)
    octF = np.ascontiguousarray(octF, np.int32)
    off, idx = (np.ascontiguousarray(a, np.uint32) for a in grid)
    uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
    bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
    q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
    qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
    assert len(qd) == len(q)
    m, b, s2 = (np.zeros(len(q), np.int32) for _ in range(3))
    is2 = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    rc = lib().orc_search_by_projection_chi2(_p(descF), _p(xyF), _p(octF), len(descF), _p(off), _p(idx), *[float(v) for v in bounds],
                                             _p(uR), _p(bl), _p(is2), 0 if is2 is None else len(is2), _p(q), _p(qd), len(q), int(th),
                                             float(nnratio), int(ratio_rule), _p(m), _p(b), _p(s2))
    assert rc == 0
    return m, b, s2


def proj_queries_last_frame(TcwC, TcwL, K, bounds_img, scale_factors, has_mp, outlier, world_pos, octL, obs_gt0, th, mono,
                            gemm_double=False):
    """the host gating of SearchByProjection(CurrentFrame, LastFrame) (:1593-1640): K = (fx, fy, cx, cy, mbf, mb),
    bounds_img = (minx, maxx, miny, maxy); returns (queries, valid)"""
    n = len(has_mp)
    q = np.zeros(n, PROJ_QUERY_DTYPE)
    valid = np.zeros(n, np.uint8)
    a = [np.ascontiguousarray(TcwC, np.float32).reshape(16), np.ascontiguousarray(TcwL, np.float32).reshape(16)]
    sf = np.ascontiguousarray(scale_factors, np.float32)
    hm, ol, og = (np.ascontiguousarray(x, np.uint8) for x in (has_mp, outlier, obs_gt0))
    wp = np.ascontiguousarray(world_pos, np.float32).reshape(-1, 3)
    oc = np.ascontiguousarray(octL, np.int32)
    rc = lib().orc_proj_queries_last_frame(_p(a[0]), _p(a[1]), *[float(v) for v in K], *[float(v) for v in bounds_img], _p(sf), n,
                                           _p(hm), _p(ol), _p(wp), _p(oc), _p(og), float(th), int(mono), int(gemm_double),
                                           _p(q), _p(valid))
    assert rc == 0
    return q, valid


def proj_queries_local_map(scale_factors, in_view, bad, scale_level, view_cos, proj_xyr, obs_gt0, th):
    n = len(in_view)
    q = np.zeros(n, PROJ_QUERY_DTYPE)
    valid = np.zeros(n, np.uint8)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    iv, bd, og = (np.ascontiguousarray(x, np.uint8) for x in (in_view, bad, obs_gt0))
    sl = np.ascontiguousarray(scale_level, np.int32)
    vc = np.ascontiguousarray(view_cos, np.float32)
    pr = np.ascontiguousarray(proj_xyr, np.float32).reshape(-1, 3)
    rc = lib().orc_proj_queries_local_map(_p(sf), n, _p(iv), _p(bd), _p(sl), _p(vc), _p(pr), _p(og), float(th), _p(q), _p(valid))
    assert rc == 0
    return q, valid


def search_for_triangulation(k1, k2, F12, ex, ey, th_low=50):
    """orc_search_for_triangulation: k1 = dict(desc, xy, elig, stereo, fv=(node, off, idx)), k2 = the same + octave,
    scale_factors, level_sigma2.  Returns match12[n1]."""
    u8, f32, i32, u32 = np.uint8, np.float32, np.int32, np.uint32
    a = [np.ascontiguousarray(k1["desc"], u8).reshape(-1, 32), np.ascontiguousarray(k1["xy"], f32).reshape(-1, 2),
         np.ascontiguousarray(k1["elig"], u8), np.ascontiguousarray(k1["stereo"], u8)]
    fv1 = [np.ascontiguousarray(x, u32) for x in k1["fv"]]
    b = [np.ascontiguousarray(k2["desc"], u8).reshape(-1, 32), np.ascontiguousarray(k2["xy"], f32).reshape(-1, 2),
         np.ascontiguousarray(k2["octave"], i32), np.ascontiguousarray(k2["elig"], u8), np.ascontiguousarray(k2["stereo"], u8)]
    fv2 = [np.ascontiguousarray(x, u32) for x in k2["fv"]]
    F = np.ascontiguousarray(F12, f32).reshape(9)
    sf, s2 = np.ascontiguousarray(k2["scale_factors"], f32), np.ascontiguousarray(k2["level_sigma2"], f32)
    m = np.full(max(len(a[0]), 1), -1, i32)
    rc = lib().orc_search_for_triangulation(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(a[0]), _p(fv1[0]), _p(fv1[1]), _p(fv1[2]), len(fv1[0]),
                                            _p(b[0]), _p(b[1]), _p(b[2]), _p(b[3]), _p(b[4]), len(b[0]), _p(fv2[0]), _p(fv2[1]), _p(fv2[2]),
                                            len(fv2[0]), _p(F), float(ex), float(ey), _p(sf), _p(s2), int(th_low), _p(m))
    assert rc == 0
    return m[:len(a[0])]
