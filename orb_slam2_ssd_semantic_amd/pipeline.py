"""Host-side mirror of the sequence pipeline of the C-ABI (include/orbfe.h orbfe_pipeline_*): the frame loop of the
reference's drivers (perfect/Examples/RGB-D/rgbd_tum.cc:77-119 -- every frame through ORBextractor::operator(), then matched
against its predecessor) for a device-resident sequence in one call.  Pointers are raw device addresses (ints); torch is
plumbing only."""
import ctypes as C

from . import _ffi
from ._ffi import OrbfeParams, check
from .extractor import ORBextractor
from .matcher import ORBmatcher


class FramePipeline:
    CONTINUE, NO_JOIN = _ffi.PIPE_CONTINUE, _ffi.PIPE_NO_JOIN

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, *, max_width=640, max_height=480,
                 sub_batch=1024, npipes=3, device=-1, blur_rounding=0, nnratio=0.9, th=100, check_ori=True):
        self._L = _ffi.lib()
        self._p = C.c_void_p()
        prm = OrbfeParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, max_width, max_height, sub_batch, device,
                          blur_rounding)
        check(self._L.orbfe_pipeline_create(C.byref(prm), npipes, C.byref(self._p)), "orbfe_pipeline_create")
        self.npipes, self.sub_batch, self.blur_rounding = npipes, sub_batch, blur_rounding
        self.cap = int(self._L.orbfe_pipeline_capacity(self._p))
        self.nnratio, self.th, self.check_ori = float(nnratio), int(th), bool(check_ori)
        kw = dict(max_width=max_width, max_height=max_height, max_batch=sub_batch, device=device, blur_rounding=blur_rounding)
        self.extractors = [ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST,
                                        _borrow=self._L.orbfe_pipeline_extractor(self._p, i), **kw) for i in range(npipes)]
        self.matchers = [ORBmatcher(nnratio, check_ori, _borrow=self._L.orbfe_pipeline_matcher(self._p, i)) for i in range(npipes)]

    def close(self):
        if getattr(self, "_p", None):
            for o in self.extractors + self.matchers:
                o.close()
            self._L.orbfe_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._p

    def capacity(self):
        return self.cap

    def extract_match_device(self, d_gray, nframes, w, h, stride, frame_stride, d_kps, d_desc, cap, d_n, d_match=None, d_nmatches=None,
                             flags=0, stream=None):
        """frames [nframes] -> padded keypoint / descriptor / count blocks (as orbfe_extract_batch_device) and, with d_match,
        row k = matches of frame k into frame k - 1 (across sub-batches; frame 0: previous call's last frame with CONTINUE)"""
        check(self._L.orbfe_pipeline_extract_match_device(self._p, d_gray, nframes, w, h, stride, frame_stride, d_kps, d_desc, cap, d_n,
                                                          d_match, d_nmatches, self.nnratio, self.th, int(self.check_ori), flags,
                                                          stream), "orbfe_pipeline_extract_match_device")

    def extract_match(self, frame_ptrs, nframes, w, h, stride, kps_ptr, desc_ptr, cap, n_ptr, match_ptr=None, nmatches_ptr=None, flags=0):
        """host frames in, host results out (orbfe_pipeline_extract_match): frame_ptrs = ctypes array of nframes host pointers;
        the other arguments are host addresses (ints).  Blocking."""
        check(self._L.orbfe_pipeline_extract_match(self._p, frame_ptrs, nframes, w, h, stride, kps_ptr, desc_ptr, cap, n_ptr, match_ptr,
                                                   nmatches_ptr, self.nnratio, self.th, int(self.check_ori), flags),
              "orbfe_pipeline_extract_match")

    def set_host_pipes(self, n):
        check(self._L.orbfe_pipeline_set_host_pipes(self._p, int(n)), "orbfe_pipeline_set_host_pipes")

    def join(self, stream=None):
        check(self._L.orbfe_pipeline_join(self._p, stream), "orbfe_pipeline_join")

    def synchronize(self):
        check(self._L.orbfe_pipeline_synchronize(self._p), "orbfe_pipeline_synchronize")

    def reset_sequence(self):
        check(self._L.orbfe_pipeline_reset_sequence(self._p), "orbfe_pipeline_reset_sequence")

    def overflow(self):
        f = C.c_int32(0)
        check(self._L.orbfe_pipeline_get_overflow(self._p, C.byref(f)), "orbfe_pipeline_get_overflow")
        return f.value

    def set_fast_mode(self, mode, collect_stats=False):
        for e in self.extractors:
            e.set_fast_mode(mode, collect_stats)

    def set_bf_kernel(self, k):
        for m in self.matchers:
            m.set_bf_kernel(k)
