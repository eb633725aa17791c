"""Host-side mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:35-116) over the C-ABI.

Same constructor arguments, same call shape (`extractor(image, mask) -> keypoints, descriptors`), same
getters (`GetLevels`, `GetScaleFactors`, ...) and the public `mvImagePyramid`, so parity tests read like
the reference's own call site `(*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)`
(reference src/Frame.cc:337-343).  All compute happens in liborbfe.so's HIP kernels.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import KP_DTYPE, OrbfeParams, check, ptr


class ORBextractor:
    HARRIS_SCORE, FAST_SCORE = 0, 1  # declared, unused by the reference too (include/ORBextractor.h:39)

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, *,
                 max_width=640, max_height=480, max_batch=1, device=-1, blur_rounding=0, options=None, lib=None, _borrow=None):
        """options: {name: value} for orbfe_set_option (names: _ffi.OPTIONS), applied before the first call.  lib: another
        build of liborbfe (_ffi.load_variant), e.g. the developer build that holds the measured-slower kernel variants."""
        self._L = lib or _ffi.lib()
        self._h = C.c_void_p()
        self._owned = _borrow is None
        if _borrow is not None:   # a handle owned by someone else (a pipe of orbfe_pipeline): never destroyed here
            self._h = C.c_void_p(_borrow)
        else:
            p = OrbfeParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height, max_batch,
                            device, blur_rounding)
            check(self._L.orbfe_create(C.byref(p), C.byref(self._h)), "orbfe_create")
        for k, v in (options or {}).items():
            self.set_option(k, v)
        self.nfeatures, self.scaleFactor, self.nlevels = nfeatures, float(np.float32(scaleFactor)), nlevels
        self.iniThFAST, self.minThFAST = iniThFAST, minThFAST
        self.max_batch = max_batch
        s = [np.zeros(nlevels, np.float32) for _ in range(4)]
        check(self._L.orbfe_get_scales(self._h, *[ptr(a) for a in s]), "orbfe_get_scales")
        self.mvScaleFactor, self.mvInvScaleFactor, self.mvLevelSigma2, self.mvInvLevelSigma2 = s
        self.mnFeaturesPerLevel = np.zeros(nlevels, np.int32)
        check(self._L.orbfe_get_features_per_level(self._h, ptr(self.mnFeaturesPerLevel)), "features_per_level")

    def close(self):
        if getattr(self, "_h", None):
            if self._owned:
                self._L.orbfe_destroy(self._h)
            self._h = None

    def set_option(self, name, value):
        """orbfe_set_option by name (overlap, rows, rows_fast, rows_blur, blur_pieces, blur_updown, pyr_rows, qt_threads_0..2,
        debug, and -- developer builds only -- pyr_fuse, fuse_blur_pyr, fuse_fast_pyr, fuse_fast_pyr_levels)"""
        check(self._L.orbfe_set_option(self._h, _ffi.OPTIONS[name], int(value)), f"orbfe_set_option({name}, {value})")

    def last_call_reused(self):
        """True when the last __call__ was answered from the previous call's results (option reuse_identical_input)"""
        return bool(self._L.orbfe_last_call_reused(self._h))

    def __del__(self):
        self.close()

    # ---- getters of include/ORBextractor.h:58-78 ----
    def GetLevels(self):
        return self.nlevels

    def GetScaleFactor(self):
        return self.scaleFactor

    def GetScaleFactors(self):
        return self.mvScaleFactor

    def GetInverseScaleFactors(self):
        return self.mvInvScaleFactor

    def GetScaleSigmaSquares(self):
        return self.mvLevelSigma2

    def GetInverseScaleSigmaSquares(self):
        return self.mvInvLevelSigma2

    @property
    def handle(self):
        return self._h

    def capacity(self):
        return int(self._L.orbfe_keypoint_capacity(self._h))

    # ---- operator() (src/ORBextractor.cc:1052-1114) ----
    def __call__(self, image, mask=None, cap=None):
        """Returns (keypoints: structured array in cv::KeyPoint field order, descriptors: N x 32 uint8).
        An empty image returns (None, None): the reference leaves its outputs untouched (:1055-1056)."""
        if image is None or image.size == 0:
            return None, None
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (:1059)"
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = cap or self.capacity()
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32(0)
        st = self._L.orbfe_extract(self._h, ptr(image), w, h, image.strides[0], ptr(kps), ptr(desc), cap,
                                   C.byref(n))
        check(st, "orbfe_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, images, cap=None):
        """Batched keyframe mode on host arrays: images = sequence of equally sized 2-D uint8 arrays."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        if not imgs:
            return []
        h, w = imgs[0].shape
        assert all(i.shape == (h, w) for i in imgs)
        nf = len(imgs)
        cap = cap or self.capacity()
        arr = (C.c_void_p * nf)(*[i.ctypes.data for i in imgs])
        kps = np.zeros((nf, cap), KP_DTYPE)
        desc = np.zeros((nf, cap, 32), np.uint8)
        n = np.zeros(nf, np.int32)
        st = self._L.orbfe_extract_batch(self._h, arr, nf, w, h, w, ptr(kps), ptr(desc), cap, ptr(n))
        check(st, "orbfe_extract_batch")
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(nf)]

    def extract_batch_device(self, d_gray, nframes, w, h, stride, frame_stride, d_kps, d_desc, cap, d_n, stream=None):
        """Raw device-pointer form (ints or ctypes pointers); asynchronous on `stream`."""
        st = self._L.orbfe_extract_batch_device(self._h, d_gray, nframes, w, h, stride, frame_stride, d_kps, d_desc,
                                                cap, d_n, stream)
        check(st, "orbfe_extract_batch_device")

    def work_counts(self):
        out = np.zeros(2, np.int64)
        check(self._L.orbfe_get_work_counts(self._h, ptr(out)), "orbfe_get_work_counts")
        return dict(fast_row_steps_per_frame=int(out[0]), fast_waves_per_frame=int(out[1]))

    def overflow(self):
        """Sticky device-side capacity flags since the last query (0 = every list fitted); waits for the last call."""
        f = C.c_int32(0)
        check(self._L.orbfe_get_overflow(self._h, C.byref(f)), "orbfe_get_overflow")
        return f.value

    def set_fast_mode(self, mode, collect_stats=False):
        check(self._L.orbfe_set_fast_mode(self._h, int(mode), int(collect_stats)), "orbfe_set_fast_mode")

    def fast_stats(self, reset=True):
        out = np.zeros(3, np.uint64)
        check(self._L.orbfe_get_fast_stats(self._h, ptr(out), int(reset)), "orbfe_get_fast_stats")
        # mode 1: {row steps, arc skips, NMS skips}; modes 2 / 3: {row steps, batches, parked pairs} of the sampled waves
        return dict(row_steps=int(out[0]), arc_skips=int(out[1]), nms_skips=int(out[2]), batches=int(out[1]), parked_pairs=int(out[2]))

    def synchronize(self):
        check(self._L.orbfe_synchronize(self._h), "orbfe_synchronize")

    # ---- mvImagePyramid (include/ORBextractor.h:80) and stage taps ----
    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        check(self._L.orbfe_get_level_size(self._h, level, C.byref(w), C.byref(h)), "orbfe_get_level_size")
        return w.value, h.value

    def pyramid_level(self, level, frame=0, with_border=False):
        w, h = self.level_size(level)
        b = 19 if with_border else 0
        out = np.zeros((h + 2 * b, w + 2 * b), np.uint8)
        check(self._L.orbfe_get_pyramid_level(self._h, frame, level, ptr(out), out.strides[0], int(with_border)),
              "orbfe_get_pyramid_level")
        return out

    def padded_pyramid(self, frame=0):
        """every level with its 19-px BORDER_REFLECT_101 frame in ONE device-to-host copy (orbfe_get_pyramid_padded): the memory
        shape of the reference's public mvImagePyramid; returns the list of (h + 38) x (w + 38) arrays"""
        off = (C.c_size_t * 16)()
        total = C.c_size_t()
        check(self._L.orbfe_get_pyramid_padded(self._h, frame, None, 0, off, C.byref(total)), "orbfe_get_pyramid_padded")
        buf = np.zeros(total.value, np.uint8)
        check(self._L.orbfe_get_pyramid_padded(self._h, frame, ptr(buf), total.value, None, None), "orbfe_get_pyramid_padded")
        out = []
        for l in range(self.nlevels):
            w, h = self.level_size(l)
            out.append(buf[off[l]:off[l] + (w + 38) * (h + 38)].reshape(h + 38, w + 38).copy())
        return out

    @property
    def mvImagePyramid(self):
        return [self.pyramid_level(l) for l in range(self.nlevels)]

    def blurred_level(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(self._L.orbfe_tap_blurred_level(self._h, frame, level, ptr(out), out.strides[0]), "tap_blurred")
        return out

    def _tap_xyr(self, fn, level, frame):
        n = C.c_int32(0)
        st = fn(self._h, frame, level, None, 0, C.byref(n))
        if st not in (_ffi.ORBFE_OK, _ffi.ORBFE_ERR_CAP):
            check(st, "tap")
        out = np.zeros((max(n.value, 1), 3), np.float32)
        check(fn(self._h, frame, level, ptr(out), n.value, C.byref(n)), "tap")
        return out[:n.value]

    def candidates(self, level, frame=0):
        return self._tap_xyr(self._L.orbfe_tap_candidates, level, frame)

    def selected(self, level, frame=0):
        return self._tap_xyr(self._L.orbfe_tap_selected, level, frame)

    # ---- timing ----
    def set_profiling(self, on=True):
        check(self._L.orbfe_set_profiling(self._h, int(on)), "orbfe_set_profiling")

    def stage_ms(self):
        ms = np.zeros(6, np.float32)
        check(self._L.orbfe_get_stage_ms(self._h, ptr(ms)), "orbfe_get_stage_ms")
        return dict(zip(_ffi.STAGES, [float(v) for v in ms]))
