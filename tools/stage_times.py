"""Stage times of the batched extractor on B distinct S(seed) frames (GPU box).
usage: B=256 [W=1920 H=1080 NF=4000] [OVERLAP=0] [ROWS_BLUR=n] [ROWS_FAST=n] [ORBFE_LIB=ab/liborbfe_x.so] python tools/stage_times.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam2_ssd_semantic_amd import ORBextractor
from bench import base_frames, expand_frames

B = int(os.environ.get("B", "256"))
GEN = os.environ.get("GEN", "S")
w, h, NF = int(os.environ.get("W", "640")), int(os.environ.get("H", "480")), int(os.environ.get("NF", "1000"))
ext = ORBextractor(NF, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
ext.set_fast_mode(int(os.environ.get("FAST_MODE", "0")))
if "OVERLAP" in os.environ:      # the release library reads no environment: the option goes through the C-ABI
    ext.set_option("overlap", int(os.environ["OVERLAP"]))
for name in ("rows_blur", "rows_fast"):
    if name.upper() in os.environ:
        ext.set_option(name, int(os.environ[name.upper()]))
cap = ext.capacity()
fr = expand_frames(torch.from_numpy(base_frames(GEN, min(B, 32), w, h, 10000)).cuda(), B)
dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
dn = torch.zeros(B, dtype=torch.int32, device="cuda")


def run(n):
    for _ in range(n):
        ext.extract_batch_device(fr.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), None)


run(3)
torch.cuda.synchronize()
ext.set_profiling(True)
run(10)
torch.cuda.synchronize()
print(GEN, B, {k: round(v, 4) for k, v in ext.stage_ms().items()})
ext.set_profiling(False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
run(3)
torch.cuda.synchronize()
e0.record()
run(20)
e1.record()
torch.cuda.synchronize()
print("total_ms_unprofiled", round(e0.elapsed_time(e1) / 20, 4), "overflow", ext.overflow())
