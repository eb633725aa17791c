"""A/B of builds of the lane-compacting FAST kernel (k_fast_map_c): the FAST stage time of one 1024-frame launch in mode 2 on S_tum
(and on S_tum with reduced contrast, and on S), per variant library, with the outputs compared against the default library's
dense mode.  usage: python tools/compact_ab.py [ab/liborbfe_x.so ...]   (the default library is always measured first)"""
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_ssd_semantic_amd import ORBextractor, _ffi  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frames_parallel  # noqa: E402

F, w, h = 1024, 640, 480
libs = [("default", None)] + [(os.path.basename(p), _ffi.load_variant(os.path.join(ROOT, p))) for p in sys.argv[1:]]
sets = []
for gen, nseed, contrast in [("S_tum", 64, float(c)) for c in os.environ.get("CONTRASTS", "1.0,0.4").split(",")] + [("S", 64, 1.0)]:
    base = torch.from_numpy(synth_frames_parallel(gen, nseed, h, w, 10000)).cuda()
    if contrast != 1.0:
        base = (128.0 + (base.float() - 128.0) * contrast).round().clamp(0, 255).to(torch.uint8)
    sets.append((gen if contrast == 1.0 else f"{gen}_c{contrast}", base.repeat((F + nseed - 1) // nseed, 1, 1)[:F].contiguous()))


def run(lib, mode, g):
    kw = {} if lib is None else {"lib": lib}
    e = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=F, options={"overlap": 0}, **kw)
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
    ms = e.stage_ms()
    dig = hashlib.sha1(n.cpu().numpy().tobytes() + k.cpu().numpy().tobytes() + d.cpu().numpy().tobytes()).hexdigest()[:12]
    e.close()
    return round(ms["fast"], 4), round(sum(v for kk, v in ms.items() if kk != "total"), 4), dig


out = {}
for name, g in sets:
    row = {}
    ms, tot, ref = run(None, 0, g)
    row["dense"] = ms
    for label, lib in libs:
        ms, tot, dig = run(lib, 2, g)
        row[label] = ms
        if dig != ref:
            row[label + "_MISMATCH"] = True
        e2 = ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=F, **({} if lib is None else {"lib": lib}))
        e2.set_fast_mode(2, collect_stats=True)
        e2.fast_stats(reset=True)
        kk = torch.zeros((F, e2.capacity(), 7), dtype=torch.int32, device="cuda")
        dd = torch.zeros((F, e2.capacity(), 32), dtype=torch.uint8, device="cuda")
        nn = torch.zeros(F, dtype=torch.int32, device="cuda")
        e2.extract_batch_device(g.data_ptr(), F, w, h, w, w * h, kk.data_ptr(), dd.data_ptr(), e2.capacity(), nn.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        fs = e2.fast_stats()
        row["pass_rate"] = round(fs["parked_pairs"] / (128.0 * max(fs["row_steps"], 1)), 4)
        row["mean_kp"] = round(float(nn.float().mean()), 1)
        e2.close()
    out[name] = row
    print(name, json.dumps(row), flush=True)
print(json.dumps(out))
