"""PCIe-inclusive leg alone (orbfe_pipeline_extract_match), for a few pipe counts / chunk sizes.  usage: python tools/pcie_probe.py"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from orb_slam2_ssd_semantic_amd.synth import synth_frames_parallel  # noqa: E402

class A: pass
out = {}
base = torch.from_numpy(synth_frames_parallel("S", 256, 480, 640, 10000)).cuda()
for pipes, F, hp in ((12, 1024, 1), (12, 1024, 2), (3, 1024, 1), (12, 2048, 1)):
    a = A(); a.width, a.height, a.no_match, a.pipes, a.blur_rounding, a.step_join = 640, 480, False, pipes, 0, False
    eng = bench.HipEngine(a, 0, 1000, F, 2, 1)
    eng.pl.set_host_pipes(hp)
    d = bench.expand_frames(base, F)
    r = bench.pcie_leg(eng, d, 640, 480, F, nbatches=max(6, 24576 // F))
    out[f"pipes{pipes}_F{F}_host{hp}"] = {k: r[k] for k in ("frames_per_s", "h2d_GBps_measured", "frac_of_link_bound")}
    eng.pl.close(); del eng; torch.cuda.empty_cache()
print(json.dumps(out))
