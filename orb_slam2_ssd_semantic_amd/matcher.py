"""Host-side mirror of the Hamming core of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:33-134).

Same constructor `(nnratio=0.6, checkOri=True)`, same public constants, same method names for the
part of the class that is on the hot path.  Map points / keyframes are passed
as plain arrays: descriptors, "has a good MapPoint" flags, keypoint angles and the DBoW2
FeatureVector as CSR (node ids ascending, offsets, feature indices).
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check, ptr


def feature_vector_to_csr(fv):
    """dict / mapping {node_id: [feature indices]} (DBoW2::FeatureVector) -> (node, off, idx) uint32 arrays."""
    nodes = sorted(fv.keys())
    node = np.asarray(nodes, np.uint32)
    off = np.zeros(len(nodes) + 1, np.uint32)
    idx = []
    for i, k in enumerate(nodes):
        idx.extend(int(v) for v in fv[k])
        off[i + 1] = len(idx)
    return node, off, np.asarray(idx, np.uint32)


class ORBmatcher:
    TH_HIGH = 100      # src/ORBmatcher.cc:39
    TH_LOW = 50        # src/ORBmatcher.cc:40
    HISTO_LENGTH = 30  # src/ORBmatcher.cc:41

    def __init__(self, nnratio=0.6, checkOri=True, device=-1, _borrow=None):
        self._L = _ffi.lib()
        self._m = C.c_void_p()
        self._owned = _borrow is None
        if _borrow is not None:   # a matcher owned by a pipeline
            self._m = C.c_void_p(_borrow)
        else:
            check(self._L.orbfe_matcher_create(device, C.byref(self._m)), "orbfe_matcher_create")
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    def close(self):
        if getattr(self, "_m", None):
            if self._owned:
                self._L.orbfe_matcher_destroy(self._m)
            self._m = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._m

    @staticmethod
    def DescriptorDistance(a, b):
        """static int DescriptorDistance(const cv::Mat&, const cv::Mat&) (src/ORBmatcher.cc:1968-1984)."""
        a = np.ascontiguousarray(a, np.uint8).reshape(32)
        b = np.ascontiguousarray(b, np.uint8).reshape(32)
        return int(_ffi.lib().orbfe_hamming(ptr(a), ptr(b)))

    def MatchBruteForce(self, descQ, descT, anglesQ=None, anglesT=None, th=None):
        """BASELINE config 3: all-pairs best / second-best + ratio + rotation histogram.
        Returns (match_q2t, best, second, nmatches)."""
        q = np.ascontiguousarray(descQ, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(descT, np.uint8).reshape(-1, 32)
        qa = None if anglesQ is None else np.ascontiguousarray(anglesQ, np.float32)
        ta = None if anglesT is None else np.ascontiguousarray(anglesT, np.float32)
        th = self.TH_HIGH if th is None else th
        m = np.full(len(q), -1, np.int32)
        b = np.full(len(q), 256, np.int32)
        s = np.full(len(q), 256, np.int32)
        n = C.c_int32(0)
        st = self._L.orbfe_match_bf(self._m, ptr(q), len(q), ptr(t), len(t), ptr(qa), ptr(ta), self.mfNNratio, th,
                                    int(self.mbCheckOrientation), ptr(m), ptr(b), ptr(s), C.byref(n))
        check(st, "orbfe_match_bf")
        return m, b, s, n.value

    def SearchByBoW(self, descKF, validKF, anglesKF, fvKF, descF, validF, anglesF, fvF, strict_lt=None, th_low=None):
        """SearchByBoW(KeyFrame*, Frame&, ...) when validF is None (src/ORBmatcher.cc:217-363),
        SearchByBoW(KeyFrame*, KeyFrame*, ...) otherwise (src/ORBmatcher.cc:665-812).
        fvKF / fvF: (node, off, idx) CSR or a {node: [indices]} mapping.  Returns (matchF2KF, nmatches)."""
        if isinstance(fvKF, dict):
            fvKF = feature_vector_to_csr(fvKF)
        if isinstance(fvF, dict):
            fvF = feature_vector_to_csr(fvF)
        dk = np.ascontiguousarray(descKF, np.uint8).reshape(-1, 32)
        df = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
        vk = None if validKF is None else np.ascontiguousarray(validKF, np.uint8)
        vf = None if validF is None else np.ascontiguousarray(validF, np.uint8)
        ak = np.ascontiguousarray(anglesKF, np.float32)
        af = np.ascontiguousarray(anglesF, np.float32)
        nk, ok, ik = [np.ascontiguousarray(a, np.uint32) for a in fvKF]
        nf, of, if_ = [np.ascontiguousarray(a, np.uint32) for a in fvF]
        if strict_lt is None:
            strict_lt = validF is not None
        th_low = self.TH_LOW if th_low is None else th_low
        m = np.full(len(df), -1, np.int32)
        n = C.c_int32(0)
        st = self._L.orbfe_search_by_bow(self._m, ptr(dk), len(dk), ptr(vk), ptr(ak), ptr(nk), ptr(ok), ptr(ik),
                                         len(nk), ptr(df), len(df), ptr(vf), ptr(af), ptr(nf), ptr(of), ptr(if_),
                                         len(nf), self.mfNNratio, th_low, int(strict_lt),
                                         int(self.mbCheckOrientation), ptr(m), C.byref(n))
        check(st, "orbfe_search_by_bow")
        return m, n.value

    def set_bf_kernel(self, kernel):
        """0 = matrix-core int8 dot product (default), 1 = xor / popcount: same results (orbfe_matcher_set_bf_kernel)"""
        check(self._L.orbfe_matcher_set_bf_kernel(self._m, int(kernel)), "orbfe_matcher_set_bf_kernel")

    def set_projection_kernel(self, kernel):
        """0 = the projection search in one launch (default), 1 = the four-kernel path: same results (orbfe_matcher_set_projection_kernel)"""
        check(self._L.orbfe_matcher_set_projection_kernel(self._m, int(kernel)), "orbfe_matcher_set_projection_kernel")

    def SearchByBoW_batch_device(self, d_kps, d_desc, cap, d_valid, d_fv_node, d_fv_off, d_fv_idx, d_counts, d_kf, d_f,
                                 npairs, d_match, d_nmatches, kf_kf=False, th_low=None, stream=None):
        """SearchByBoW for a batch of (KeyFrame, Frame) pairs taken from device-resident extractor / BoW blocks
        (orbfe_search_by_bow_batch_device); device pointers (ints), asynchronous on `stream`."""
        th_low = self.TH_LOW if th_low is None else th_low
        check(self._L.orbfe_search_by_bow_batch_device(self._m, d_kps, d_desc, cap, d_valid, d_fv_node, d_fv_off, d_fv_idx,
                                                       d_counts, d_kf, d_f, npairs, self.mfNNratio, th_low, int(kf_kf),
                                                       int(self.mbCheckOrientation), d_match, d_nmatches, stream),
              "orbfe_search_by_bow_batch_device")

    def HammingCSR(self, descQ, descT, off, cand):
        """best / second-best over per-query candidate lists (SearchByProjection family)."""
        q = np.ascontiguousarray(descQ, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(descT, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, np.uint32)
        cand = np.ascontiguousarray(cand, np.uint32)
        bi = np.full(len(q), -1, np.int32)
        b = np.full(len(q), 256, np.int32)
        s = np.full(len(q), 256, np.int32)
        st = self._L.orbfe_hamming_csr(self._m, ptr(q), len(q), ptr(t), len(t), ptr(off), ptr(cand), ptr(bi), ptr(b),
                                       ptr(s))
        check(st, "orbfe_hamming_csr")
        return bi, b, s

    def HammingCSRAll(self, descQ, descT, off, cand):
        """Every distance of every candidate list (orbfe_hamming_csr_all): dist[off[-1]] uint16."""
        q = np.ascontiguousarray(descQ, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(descT, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, np.uint32)
        cand = np.ascontiguousarray(cand, np.uint32)
        d = np.zeros(int(off[-1]) if len(off) else 0, np.uint16)
        check(self._L.orbfe_hamming_csr_all(self._m, ptr(q), len(q), ptr(t), len(t), ptr(off), ptr(cand), ptr(d)), "orbfe_hamming_csr_all")
        return d

    def HammingCSR2(self, descQ, descT, off, cand):
        """HammingCSR plus second_idx: the candidate owning the runner-up distance (orbfe_hamming_csr_ex)."""
        q = np.ascontiguousarray(descQ, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(descT, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, np.uint32)
        cand = np.ascontiguousarray(cand, np.uint32)
        bi, si = np.full(len(q), -1, np.int32), np.full(len(q), -1, np.int32)
        b, s = np.full(len(q), 256, np.int32), np.full(len(q), 256, np.int32)
        check(self._L.orbfe_hamming_csr_ex(self._m, ptr(q), len(q), ptr(t), len(t), ptr(off), ptr(cand), ptr(bi), ptr(b), ptr(s),
                                           ptr(si)), "orbfe_hamming_csr_ex")
        return bi, b, s, si

    def SearchByProjectionCore(self, descF, xyF, octF, grid, bounds, uRight, blocked, queries, qdesc, th, nnratio, ratio_rule,
                               inv_level_sigma2=None):
        """orbfe_search_by_projection: the device core of ORBmatcher::SearchByProjection (Frame&, const Frame&, th, bMono)
        (src/ORBmatcher.cc:1578-1724) and (Frame&, const vector<MapPoint*>&, th) (:63-157).  grid = (cell_off, cell_idx) of
        the frame, bounds = (minx, miny, gw_inv, gh_inv), queries: PROJ_QUERY_DTYPE records, one per MapPoint that passed the
        host-side gates.  Returns (match[nq], best[nq], second[nq])."""
        descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
        xyF = np.ascontiguousarray(xyF, np.float32).reshape(-1, 2)
        octF = np.ascontiguousarray(octF, np.int32)
        off, idx = (np.ascontiguousarray(a, np.uint32) for a in grid)
        uR = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        bl = None if blocked is None else np.ascontiguousarray(blocked, np.uint8)
        q = np.ascontiguousarray(queries, _ffi.PROJ_QUERY_DTYPE)
        qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        assert len(qd) == len(q) and len(xyF) == len(descF) == len(octF)
        m, b, s2 = (np.full(len(q), -1, np.int32) for _ in range(3))
        if inv_level_sigma2 is not None:   # queries may carry ORBFE_PROJ_CHI2_GATE (Fuse, src/ORBmatcher.cc:1112-1139)
            is2 = np.ascontiguousarray(inv_level_sigma2, np.float32)
            _ffi.check(_ffi.lib().orbfe_search_by_projection_chi2(
                self.handle, _ffi.ptr(descF), _ffi.ptr(xyF), _ffi.ptr(octF), len(descF), _ffi.ptr(off), _ffi.ptr(idx),
                *[float(v) for v in bounds], _ffi.ptr(uR), _ffi.ptr(bl), _ffi.ptr(is2), len(is2), _ffi.ptr(q), _ffi.ptr(qd), len(q),
                int(th), float(nnratio), int(ratio_rule), _ffi.ptr(m), _ffi.ptr(b), _ffi.ptr(s2)), "orbfe_search_by_projection_chi2")
            return m, b, s2
        _ffi.check(_ffi.lib().orbfe_search_by_projection(self.handle, _ffi.ptr(descF), _ffi.ptr(xyF), _ffi.ptr(octF), len(descF),
                                                         _ffi.ptr(off), _ffi.ptr(idx), *[float(v) for v in bounds], _ffi.ptr(uR),
                                                         _ffi.ptr(bl), _ffi.ptr(q), _ffi.ptr(qd), len(q), int(th), float(nnratio),
                                                         int(ratio_rule), _ffi.ptr(m), _ffi.ptr(b), _ffi.ptr(s2)),
                   "orbfe_search_by_projection")
        return m, b, s2

    def WindowDistances(self, descF, xyF, octF, grid, bounds, queries, qdesc, cap=None):
        """orbfe_window_distances: per query the features of GetFeaturesInArea(u, v, r, min_level, max_level) in the
        reference's order with their Hamming distances.  Returns (off[nq + 1], cand[], dist[])."""
        descF = np.ascontiguousarray(descF, np.uint8).reshape(-1, 32)
        xyF = np.ascontiguousarray(xyF, np.float32).reshape(-1, 2)
        octF = np.ascontiguousarray(octF, np.int32)
        goff, gidx = (np.ascontiguousarray(a, np.uint32) for a in grid)
        q = np.ascontiguousarray(queries, _ffi.PROJ_QUERY_DTYPE)
        qd = np.ascontiguousarray(qdesc, np.uint8).reshape(-1, 32)
        off = np.zeros(len(q) + 1, np.uint32)
        cap = int(cap if cap is not None else max(64 * len(q), 1024))
        for _ in range(2):
            ent = np.zeros(max(cap, 1), np.uint32)
            st = _ffi.lib().orbfe_window_distances(self.handle, _ffi.ptr(descF), _ffi.ptr(xyF), _ffi.ptr(octF), len(descF), _ffi.ptr(goff),
                                                   _ffi.ptr(gidx), *[float(v) for v in bounds], _ffi.ptr(q), _ffi.ptr(qd), len(q),
                                                   _ffi.ptr(off), _ffi.ptr(ent), cap)
            if st == _ffi.ORBFE_ERR_CAP:
                cap = int(off[-1])
                continue
            _ffi.check(st, "orbfe_window_distances")
            break
        n = int(off[-1])
        return off, ent[:n] & 0xFFFF, ent[:n] >> 16

    def SearchForTriangulationCore(self, k1, k2, F12, ex, ey, th_low=50):
        """orbfe_search_for_triangulation (src/ORBmatcher.cc:827-1012): k1 = dict(desc, xy, elig, stereo, fv=(node, off, idx)),
        k2 = the same + octave, scale_factors, level_sigma2.  Returns match12[n1] (before the rotation check)."""
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
        _ffi.check(_ffi.lib().orbfe_search_for_triangulation(
            self.handle, _ffi.ptr(a[0]), _ffi.ptr(a[1]), _ffi.ptr(a[2]), _ffi.ptr(a[3]), len(a[0]), _ffi.ptr(fv1[0]), _ffi.ptr(fv1[1]),
            _ffi.ptr(fv1[2]), len(fv1[0]), _ffi.ptr(b[0]), _ffi.ptr(b[1]), _ffi.ptr(b[2]), _ffi.ptr(b[3]), _ffi.ptr(b[4]), len(b[0]),
            _ffi.ptr(fv2[0]), _ffi.ptr(fv2[1]), _ffi.ptr(fv2[2]), len(fv2[0]), _ffi.ptr(F), float(ex), float(ey), _ffi.ptr(sf), _ffi.ptr(s2),
            len(sf), int(th_low), _ffi.ptr(m)), "orbfe_search_for_triangulation")
        return m[:len(a[0])]

    def ComputeStereoMatches(self, extractorLeft, extractorRight, keysL, descL, keysR, descR, mbf, mb):
        """Frame::ComputeStereoMatches (src/Frame.cc:642-846).  The two ORBextractor mirrors must have
        just processed the left / right image (their device-resident pyramids are read).  Returns (mvuRight, mvDepth)."""
        from ._ffi import KP_DTYPE
        kl = np.ascontiguousarray(keysL, KP_DTYPE)
        kr = np.ascontiguousarray(keysR, KP_DTYPE)
        dl = np.ascontiguousarray(descL, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(descR, np.uint8).reshape(-1, 32)
        n = len(kl)
        u = np.full(max(n, 1), -1, np.float32)
        d = np.full(max(n, 1), -1, np.float32)
        st = self._L.orbfe_stereo_matches(self._m, extractorLeft.handle, extractorRight.handle, ptr(kl), ptr(dl), n,
                                          ptr(kr), ptr(dr), len(kr), float(mbf), float(mb), ptr(u), ptr(d))
        check(st, "orbfe_stereo_matches")
        return u[:n].copy(), d[:n].copy()

    def ComputeDistinctiveDescriptors_device(self, d_pool, d_off, d_idx, npoints, max_obs, d_best, d_median, stream=None):
        """MapPoint::ComputeDistinctiveDescriptors on device buffers (pointers as ints); -2 marks points beyond max_obs."""
        check(self._L.orbfe_distinctive_descriptors_device(self._m, d_pool, d_off, d_idx, int(npoints), int(max_obs), d_best, d_median,
                                                           stream), "orbfe_distinctive_descriptors_device")

    def AssignFeaturesToGrid_batch_device(self, d_kps, d_n, cap, nframes, minx, miny, gw_inv, gh_inv, d_cell_off, d_cell_idx,
                                          d_n_in_grid, stream=None):
        """Frame::AssignFeaturesToGrid for every frame of an extractor output block (device pointers as ints)."""
        check(self._L.orbfe_assign_grid_batch_device(self._m, d_kps, d_n, int(cap), int(nframes), float(minx), float(miny),
                                                     float(gw_inv), float(gh_inv), d_cell_off, d_cell_idx, d_n_in_grid, stream),
              "orbfe_assign_grid_batch_device")

    def GetFeaturesInArea_device(self, d_kps, d_cell_off, d_cell_idx, minx, miny, gw_inv, gh_inv, d_qxyr, d_qlevels, nq, d_off, d_cand,
                                 cap, stream=None):
        """Frame::GetFeaturesInArea for nq queries on one frame of an extractor output block (device pointers as ints)."""
        check(self._L.orbfe_features_in_area_device(self._m, d_kps, d_cell_off, d_cell_idx, float(minx), float(miny), float(gw_inv),
                                                    float(gh_inv), d_qxyr, d_qlevels, int(nq), d_off, d_cand, int(cap), stream),
              "orbfe_features_in_area_device")

    def ComputeStereoMatches_batch_device(self, extractorLeft, extractorRight, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, cap,
                                          nframes, mbf, mb, d_uRight, d_depth, stream=None):
        """Frame::ComputeStereoMatches for every frame pair of the two extractors' last device batches; all arguments are
        device pointers (ints) to the blocks those calls wrote; results [nframes][cap] floats on `stream`."""
        st = self._L.orbfe_stereo_matches_batch_device(self._m, extractorLeft.handle, extractorRight.handle, d_kpsL, d_descL, d_nL,
                                                       d_kpsR, d_descR, d_nR, int(cap), int(nframes), float(mbf), float(mb),
                                                       d_uRight, d_depth, stream)
        check(st, "orbfe_stereo_matches_batch_device")

    def ComputeDistinctiveDescriptors(self, pool, off, idx):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:284-345) for a batch of map points.
        pool: descriptors; map point p observes pool[idx[off[p]:off[p+1]]].  Returns (best position inside the
        point's list, its median distance), -1 / -1 for points without observations."""
        pool = np.ascontiguousarray(pool, np.uint8).reshape(-1, 32)
        off = np.ascontiguousarray(off, np.uint32)
        idx = np.ascontiguousarray(idx, np.uint32)
        npts = len(off) - 1
        best = np.full(max(npts, 1), -1, np.int32)
        med = np.full(max(npts, 1), -1, np.int32)
        st = self._L.orbfe_distinctive_descriptors(self._m, ptr(pool), len(pool), ptr(off), ptr(idx), npts, ptr(best),
                                                   ptr(med))
        check(st, "orbfe_distinctive_descriptors")
        return best[:npts].copy(), med[:npts].copy()


class FrameGrid:
    """Device-built mGrid of a Frame (reference src/Frame.cc:319-334) + batched GetFeaturesInArea (:465-518).

    `FrameGrid(matcher, keysUn_xy, octaves, minX, minY, gridElementWidthInv, gridElementHeightInv)`; then
    `GetFeaturesInArea(x, y, r, minLevel=-1, maxLevel=-1)` for one query or `query(qxyr, qlevels)` for a batch
    (CSR result to feed `ORBmatcher.HammingCSR`)."""
    COLS, ROWS = 64, 48

    def __init__(self, matcher, xy, octave, minx, miny, gw_inv, gh_inv):
        self._mt = matcher
        self.xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.p = (float(np.float32(minx)), float(np.float32(miny)), float(np.float32(gw_inv)), float(np.float32(gh_inv)))
        n = len(self.xy)
        self.cell_off = np.zeros(self.COLS * self.ROWS + 1, np.uint32)
        self.cell_idx = np.zeros(max(n, 1), np.uint32)
        nin = C.c_int32(0)
        st = matcher._L.orbfe_assign_grid(matcher._m, ptr(self.xy), n, *self.p, ptr(self.cell_off), ptr(self.cell_idx),
                                          C.byref(nin))
        check(st, "orbfe_assign_grid")
        self.cell_idx = self.cell_idx[:nin.value].copy()

    def mGrid(self, ix, iy):
        c = ix * self.ROWS + iy
        return self.cell_idx[self.cell_off[c]:self.cell_off[c + 1]]

    def query(self, qxyr, qlevels=None, cap=None):
        q = np.ascontiguousarray(qxyr, np.float32).reshape(-1, 3)
        ql = None if qlevels is None else np.ascontiguousarray(qlevels, np.int32).reshape(-1, 2)
        nq = len(q)
        off = np.zeros(nq + 1, np.uint32)
        cap = cap or max(64 * nq, 1024)
        while True:
            cand = np.zeros(max(cap, 1), np.uint32)
            st = self._mt._L.orbfe_features_in_area(self._mt._m, ptr(self.xy), ptr(self.octave), len(self.xy),
                                                    ptr(self.cell_off), ptr(self.cell_idx if len(self.cell_idx) else
                                                                            np.zeros(1, np.uint32)), *self.p,
                                                    ptr(q), ptr(ql), nq, ptr(off), ptr(cand), cap)
            if st == _ffi.ORBFE_ERR_CAP:
                cap = int(off[nq])
                continue
            check(st, "orbfe_features_in_area")
            return off, cand[:off[nq]].copy()

    def GetFeaturesInArea(self, x, y, r, minLevel=-1, maxLevel=-1):
        off, cand = self.query([[x, y, r]], [[minLevel, maxLevel]])
        return cand


class ORBVocabulary:
    """DBoW2 vocabulary tree on the device (reference include/ORBVocabulary.h).

    `ORBVocabulary(matcher, child_off, child_idx, node_desc, word_id, weight, L)`; `transform(descriptors, levelsup=4)`
    mirrors `mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)` (src/Frame.cc:553) and returns
    (BowVector as (ids, values), FeatureVector as the (node, off, idx) CSR `ORBmatcher.SearchByBoW` takes)."""

    def __init__(self, matcher, child_off, child_idx, node_desc, word_id, weight, L, device=-1):
        self._mt = matcher
        self._L = _ffi.lib()
        self._v = C.c_void_p()
        co = np.ascontiguousarray(child_off, np.uint32)
        ci = np.ascontiguousarray(child_idx, np.uint32)
        nd = np.ascontiguousarray(node_desc, np.uint8).reshape(-1, 32)
        wi = np.ascontiguousarray(word_id, np.uint32)
        ww = np.ascontiguousarray(weight, np.float64)
        check(self._L.orbfe_vocabulary_create(device, len(nd), ptr(co), ptr(ci), ptr(nd), ptr(wi), ptr(ww), int(L),
                                              C.byref(self._v)), "orbfe_vocabulary_create")

    def close(self):
        if getattr(self, "_v", None):
            self._L.orbfe_vocabulary_destroy(self._v)
            self._v = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._v

    def transform_batch_device(self, d_desc, d_n, nframes, cap, levelsup, d_f_word, d_f_node, d_f_weight, d_bow_id,
                               d_bow_val, d_fv_node, d_fv_off, d_fv_idx, d_counts, stream=None):
        """Frame::ComputeBoW for every frame of an extractor output block, device pointers (ints), asynchronous on
        `stream`: see orbfe_bow_transform_batch_device in include/orbfe.h for the layouts."""
        check(self._L.orbfe_bow_transform_batch_device(self._mt._m, self._v, d_desc, d_n, nframes, cap, levelsup, d_f_word,
                                                       d_f_node, d_f_weight, d_bow_id, d_bow_val, d_fv_node, d_fv_off,
                                                       d_fv_idx, d_counts, stream), "orbfe_bow_transform_batch_device")

    def transform(self, desc, levelsup=4, per_feature=False):
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        m = max(n, 1)
        fw, fn, fwt = np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros(m, np.float64)
        bid, bval, nb = np.zeros(m, np.uint32), np.zeros(m, np.float64), C.c_int32(0)
        fvn, fvo, fvi, nf = np.zeros(m, np.uint32), np.zeros(m + 1, np.uint32), np.zeros(m, np.uint32), C.c_int32(0)
        st = self._L.orbfe_bow_transform(self._mt._m, self._v, ptr(d), n, levelsup, ptr(fw), ptr(fn), ptr(fwt), ptr(bid),
                                         ptr(bval), C.byref(nb), ptr(fvn), ptr(fvo), ptr(fvi), C.byref(nf))
        check(st, "orbfe_bow_transform")
        bow = (bid[:nb.value].copy(), bval[:nb.value].copy())
        fv = (fvn[:nf.value].copy(), fvo[:nf.value + 1].copy(), fvi[:int(fvo[nf.value])].copy())
        if per_feature:
            return bow, fv, (fw[:n].copy(), fn[:n].copy(), fwt[:n].copy())
        return bow, fv
