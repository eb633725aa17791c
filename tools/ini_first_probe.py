"""Would the reference's own FAST order -- cv::FAST(iniThFAST) on every cell, cv::FAST(minThFAST) only on the cells that came back
empty (src/ORBextractor.cc:818-825) -- beat the shipped single pass at minThFAST?  (GPU box.)

Pass 1 of that order is exactly what the shipped kernels do when minThFAST == iniThFAST: the lane-compacting kernel's necessary
test and survivor threshold both run at 20.  So its cost is MEASURED here (FAST stage of an extractor created with (20, 20)),
for both kernel forms, next to the shipped single pass (20, 7).  Pass 2 is bounded from below: the fallback cells' tiles
(cell + 6) cover `area_fallback` of the detection window (tools/fast_pass_stats.py, CPU) and a pass over them costs at least that
share of the lane-compacting kernel's floor (0.74 ms per 1024 frames at a 2 % pass rate, profiles/r05_fast_floor_v2.json).
    two_pass_lower_bound = fast_ms(20, 20) + area_fallback * 0.74
Workloads: S, S_tum, and the 32 real-photograph VGA frames tiled to a 1024-frame batch.
usage: python tools/ini_first_probe.py > gpurun_out/r06_ini_first_probe.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_ssd_semantic_amd import ORBextractor, photos  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frames_parallel  # noqa: E402

F, w, h = 1024, 640, 480
FLOOR_MS = 0.74
cpu = json.load(open(os.path.join(ROOT, "profiles", "r06_fast_pass_stats.json")))["summary"]
sets = {
    "S": torch.from_numpy(synth_frames_parallel("S", 64, h, w, 10000)),
    "S_tum": torch.from_numpy(synth_frames_parallel("S_tum", 64, h, w, 10000)),
    "photos": torch.from_numpy(np.stack([g for _, g in photos.vga_gray_frames()])),
}
out = {}
for name, base in sets.items():
    nb = base.shape[0]
    g = base.cuda().repeat((F + nb - 1) // nb, 1, 1)[:F].contiguous()
    row = {}
    for label, mn in (("single_pass_20_7", 7), ("pass1_20_20", 20)):
        for form, mode in (("dense", 0), ("compact", 2)):
            e = ORBextractor(1000, 1.2, 8, 20, mn, max_width=w, max_height=h, max_batch=F, options={"overlap": 0})
            e.set_fast_mode(mode, collect_stats=(mode == 2))
            cap = e.capacity()
            k = torch.zeros((F, cap, 7), dtype=torch.int32, device="cuda")
            d = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
            n = torch.zeros(F, dtype=torch.int32, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
            torch.cuda.synchronize()
            if mode == 2:
                fs = e.fast_stats()
                row[f"{label}_pass_rate"] = round(fs["parked_pairs"] / (128.0 * max(fs["row_steps"], 1)), 4)
                e.set_fast_mode(2, collect_stats=False)
            e.set_profiling(True)
            for _ in range(6):
                e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
            torch.cuda.synchronize()
            row[f"{label}_{form}_fast_ms"] = round(e.stage_ms()["fast"], 4)
            e.close()
    a = cpu[name]["area_fallback"]
    row["cells_fallback_frac_cpu"] = round(cpu[name]["cells_fallback"] / cpu[name]["cells"], 4)
    row["area_fallback_cpu"] = round(a, 4)
    best1 = min(row["pass1_20_20_dense_fast_ms"], row["pass1_20_20_compact_fast_ms"])
    row["two_pass_lower_bound_ms"] = round(best1 + a * FLOOR_MS, 4)
    row["shipped_best_ms"] = min(row["single_pass_20_7_dense_fast_ms"], row["single_pass_20_7_compact_fast_ms"])
    row["two_pass_could_win"] = row["two_pass_lower_bound_ms"] < row["shipped_best_ms"]
    out[name] = row
print(json.dumps({"what": __doc__.split("\n\n")[0], "per": f"launch of {F} frames, ms", "floor_ms": FLOOR_MS, "workloads": out}))
