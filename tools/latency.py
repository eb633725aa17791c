import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam2_ssd_semantic_amd import ORBextractor
from orb_slam2_ssd_semantic_amd.synth import synth_frame
e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
img = synth_frame(1)
for _ in range(20): e(img)
lat = []
for _ in range(200):
    t = time.perf_counter(); e(img); lat.append(time.perf_counter() - t)
print("graph", os.environ.get("ORBFE_GRAPH", "1"), "median_ms", round(float(np.median(lat)) * 1e3, 4), "p90", round(float(np.percentile(lat, 90)) * 1e3, 4))
