"""Where the time of k_octree goes inside a workgroup (developer tool, GPU box): needs a build with -DQT_PROFILE
(ORBFE_EXTRA_FLAGS=-DQT_PROFILE python -c "from orb_slam2_ssd_semantic_amd import _build; _build.build(force=True)").
Prints, for the first level of every launch group and for the other levels, the mean wall-clock time per workgroup spent in
each phase.  usage: B=256 [W H NF GEN] python tools/octree_phases.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam2_ssd_semantic_amd import ORBextractor, _ffi
from bench import base_frames, expand_frames

B = int(os.environ.get("B", "256"))
GEN = os.environ.get("GEN", "S")
w, h, NF = int(os.environ.get("W", "640")), int(os.environ.get("H", "480")), int(os.environ.get("NF", "1000"))
ext = ORBextractor(NF, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
cap = ext.capacity()
fr = expand_frames(torch.from_numpy(base_frames(GEN, min(B, 32), w, h, 10000)).cuda(), B)
dk = torch.zeros((B, cap, 7), dtype=torch.int32, device="cuda")
dd = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
dn = torch.zeros(B, dtype=torch.int32, device="cuda")
L = _ffi.lib()
L.orbfe_internal_read_misc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
buf = np.zeros(128, np.uint64)
reps = 5
for it in range(2):
    for _ in range(reps):
        ext.extract_batch_device(fr.data_ptr(), B, w, h, w, w * h, dk.data_ptr(), dd.data_ptr(), cap, dn.data_ptr(), None)
    torch.cuda.synchronize()
    L.orbfe_internal_read_misc(ext.handle, buf.ctypes.data, 1)
names = ["setup", "cell flags (keys)", "histogram (keys)", "node passes", "table + flatten", "deep passes + clear", "select (keys)", "output"]
tick = 1e-8  # wall_clock64: 100 MHz
print("us per workgroup (one workgroup = one level of one frame)")
print(" " * 26 + "".join(f"   L{l}    " for l in range(8)))
tab = np.stack([buf[8 + 8 * l: 16 + 8 * l].astype(np.float64) * tick * 1e6 / (reps * B) for l in range(8)], 1)
for n, row in zip(names, tab):
    print(f"{n:26s}" + "".join(f"{x:9.2f}" for x in row))
print(f"{'total':26s}" + "".join(f"{x:9.2f}" for x in tab.sum(0)))
