"""Where the single-frame host-to-host latency goes (GPU box): total wall time of ORBextractor::operator() on one 640x480 frame
against the sum of its stage event intervals.  usage: python tools/latency_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import base_frames

w, h = int(os.environ.get("W", "640")), int(os.environ.get("H", "480"))
img = base_frames(os.environ.get("GEN", "S"), 1, w, h, 10000)[0].copy()
e = ORBextractor(int(os.environ.get("NF", "1000")), 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=1)
for _ in range(20):
    e(img)
lat = []
for _ in range(200):
    t = time.perf_counter()
    e(img)
    lat.append(time.perf_counter() - t)
e.set_profiling(True)
for _ in range(50):
    e(img)
st = e.stage_ms()
e.set_profiling(False)
print("host-to-host ms: median %.4f  p10 %.4f  p90 %.4f" % (np.median(lat) * 1e3, np.percentile(lat, 10) * 1e3, np.percentile(lat, 90) * 1e3))
print("stage event intervals (ms):", {k: round(v, 4) for k, v in st.items()})

# The C++ shim itself, called the way the reference's Frame calls it (oracle/_ref/libshim_ext.so = shim/ORBextractor.cc +
# the reference's sliced Frame::ExtractORB): with the reference's semantics (mvImagePyramid refreshed by every call, the
# shim's default) and with mbKeepPyramid = false (mono / RGB-D opt-out: no pyramid download).
try:
    from oracle import ref_ffi as R
    s = R.ShimExtractor(int(os.environ.get("NF", "1000")), 1.2, 8, 20, 7)
    for keep in (True, False):
        for _ in range(20):
            s.extract_via_frame(img, keep_pyramid=keep)
        lat = []
        for _ in range(200):
            t = time.perf_counter()
            s.extract_via_frame(img, keep_pyramid=keep)
            lat.append(time.perf_counter() - t)
        print("C++ shim operator() via Frame::ExtractORB, mbKeepPyramid = %-5s: median %.4f ms  p10 %.4f  p90 %.4f"
              % (keep, np.median(lat) * 1e3, np.percentile(lat, 10) * 1e3, np.percentile(lat, 90) * 1e3))
except Exception as ex:  # the shim libraries are test infrastructure: absent -> only the ctypes figure above
    print("C++ shim probe skipped:", ex)
