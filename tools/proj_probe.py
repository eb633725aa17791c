"""Latency of one orbfe_search_by_projection call (tracker-shaped case), for rocprofv3 --kernel-trace --stats (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import proj_cases as PC
from oracle import oracle_ffi as O
from orb_slam2_ssd_semantic_amd import ORBmatcher

mat = ORBmatcher(0.9, True)
rng = np.random.default_rng(2026)
cur, last = PC.last_frame_case(rng, 1000, 1000, "small")
q, valid = O.proj_queries_last_frame(cur["Tcw"], last["Tcw"], cur["K"], cur["bounds"], cur["scale_factors"], last["has_mp"],
                                     last["outlier"], last["world_pos"], last["octave"], last["obs_gt0"], 15.0, False)
sel = valid.astype(bool)
ci = PC.core_inputs(cur)
qq, qd = q[sel], last["mpdesc"][sel]
res = {}
for kernel, name in ((0, "one-launch (k_proj_fused)"), (1, "four-kernel"), (0, "one-launch (k_proj_fused) again")):
    mat.set_projection_kernel(kernel)
    for _ in range(5):
        out = mat.SearchByProjectionCore(queries=qq, qdesc=qd, th=100, nnratio=0.0, ratio_rule=0, **ci)
    t = []
    for _ in range(200):
        t0 = time.perf_counter()
        mat.SearchByProjectionCore(queries=qq, qdesc=qd, th=100, nnratio=0.0, ratio_rule=0, **ci)
        t.append(time.perf_counter() - t0)
    res[name] = out
    print(name, "queries", len(qq), "median call ms", round(float(np.median(t)) * 1e3, 4), "min", round(min(t) * 1e3, 4))
assert all(np.array_equal(a, b) for a, b in zip(res["one-launch (k_proj_fused)"], res["four-kernel"]))
