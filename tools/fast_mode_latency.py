"""Single-frame (and small-batch) latency of the FAST forms: host-to-host wall time of one extraction call per mode (0 dense,
2 lane-compacting) on S and S_tum frames.  usage: python tools/fast_mode_latency.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import base_frames

w, h = 640, 480
out = {}
for gen in ("S", "S_tum"):
    imgs = base_frames(gen, 8, w, h, 10000)
    for B in (1, 8):
        for mode in (0, 2):
            e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
            e.set_fast_mode(mode)
            call = (lambda: e(imgs[0])) if B == 1 else (lambda: e.extract_batch(list(imgs[:B])))
            for _ in range(20):
                call()
            lat = []
            for _ in range(150):
                t = time.perf_counter()
                call()
                lat.append(time.perf_counter() - t)
            e.set_profiling(True)
            for _ in range(30):
                call()
            st = e.stage_ms()
            e.close()
            out[f"{gen}_B{B}_mode{mode}"] = {"median_ms": round(float(np.median(lat)) * 1e3, 4), "fast_ms": round(st["fast"], 4)}
            print(gen, B, mode, out[f"{gen}_B{B}_mode{mode}"], flush=True)
print(json.dumps(out))
