"""Developer probe (needs ab/liborbfe_dev.so, ORBFE_LIB unset): what would a lane-compacting FAST cost at best?
Stage times of one 1024-frame launch on S and S_tum for: dense; sparse (dense + the 4-point compass test in front); and the
TIMING-ONLY variant ORBFE_OPT_DEBUG = 60 of the developer build, which runs unpack + compass + NMS + emission but never the
arc evaluation (results are wrong by construction).  floor = t_noarcs + pass_rate * (t_sparse - t_noarcs)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_ssd_semantic_amd import ORBextractor, _ffi  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frames_parallel  # noqa: E402

F, w, h = 1024, 640, 480
dev = _ffi.load_variant(os.path.join(ROOT, "ab", "liborbfe_dev.so"))
out = {}
# S, S_tum, and S_tum with its contrast reduced about the mean (fewer pixel pairs pass the necessary test at the fixed minTh = 7)
for gen, nseed, contrast in (("S", 128, 1.0), ("S_tum", 64, 1.0), ("S_tum", 64, 0.6), ("S_tum", 64, 0.4), ("S_tum", 64, 0.25), ("S_tum", 64, 0.15)):
    base = torch.from_numpy(synth_frames_parallel(gen, nseed, h, w, 10000)).cuda()
    if contrast != 1.0:
        base = (128.0 + (base.float() - 128.0) * contrast).round().clamp(0, 255).to(torch.uint8)
    g = base.repeat((F + nseed - 1) // nseed, 1, 1)[:F].contiguous()
    row = {}
    variants = (("dense", 0, 0), ("sparse", 1, 0), ("noarcs", 1, 60), ("compact", 2, 0)) if contrast == 1.0 else (("dense", 0, 0), ("compact", 2, 0))
    for label, mode, dbg in variants:
        e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=F, lib=dev, options={"overlap": 0, "debug": dbg})
        e.set_fast_mode(mode)
        cap = e.capacity()
        k = torch.zeros((F, cap, 7), dtype=torch.int32, device="cuda")
        d = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
        n = torch.zeros(F, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
        torch.cuda.synchronize()
        e.set_profiling(True)
        for _ in range(6):
            e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
        torch.cuda.synchronize()
        row[label] = round(e.stage_ms()["fast"], 4)
        if mode == 2:
            e.set_fast_mode(2, collect_stats=True)
            e.fast_stats(reset=True)
            e.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, k.data_ptr(), d.data_ptr(), cap, n.data_ptr(), st)
            torch.cuda.synchronize()
            fs = e.fast_stats()
            row["pass_rate"] = round(fs["parked_pairs"] / (128.0 * max(fs["row_steps"], 1)), 4)
            row["batch_fill"] = round(fs["parked_pairs"] / (64.0 * max(fs["batches"], 1)), 4)
        e.close()
    out[gen if contrast == 1.0 else f"{gen}_contrast{contrast}"] = row
print(json.dumps(out))
