"""Per-kernel roofline table of a round from the committed profiles: time (rocprofv3 kernel stats), corrected HBM traffic
(2 x FETCH_SIZE + WRITE_SIZE), achieved bandwidth against the datasheet peak and against the measured device-copy rate,
VALUBusy and wave-level VALU instruction counts.  usage: python tools/roofline_report.py r02_v6 > profiles/r02_v6_roofline.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02_v6"


def kname(n):
    return n.replace("void ", "").split("(")[0].split("<")[0]


stats = {}
for r in csv.DictReader(open(os.path.join(P, f"{tag}_kernel_stats.csv"))):
    k = kname(r["Name"])
    if k.startswith("k_"):
        stats[k] = (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"]))
hbm = json.load(open(os.path.join(P, f"{tag}_pmc_hbm.json")))
vb = json.load(open(os.path.join(P, f"{tag}_pmc_valubusy.json")))
rate = json.load(open(os.path.join(P, "r02_hbm_rate.json")))
copy = rate["copy_read_plus_write_TBps"]
F = hbm.get("frames_per_launch") or 1024
nbatch = stats["k_fast_map"][0]  # batched extractor calls in the profiled command (k_fast_map runs once per call)
print(f"# Per-kernel roofline, {tag} ({F} frames per batched call, {nbatch} calls profiled)\n")
print(f"HBM datasheet peak 8.0 TB/s; device copy measured {copy} TB/s, read {rate['read_only_sum_TBps']}, write "
      f"{rate['write_only_fill_TBps']} (profiles/r02_hbm_rate.json).  Traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 "
      "correction, profiles/r01_fetch_calibration.txt).\n")
print("| kernel | launches per call | ms per call | HBM traffic per call (GB) | achieved TB/s | of 8 TB/s | of copy rate | VALUBusy % | wave VALU instr per call (M) |")
print("|---|---|---|---|---|---|---|---|---|")
tot_ms = 0.0
for k, (calls, avg, total) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    per_call = calls / nbatch
    ms = total / nbatch / 1e6
    tot_ms += ms
    tb = hbm["corrected_hbm_bytes_per_launch"].get(k)
    gb = tb * per_call / 1e9 if tb else None
    bw = gb / ms if gb else None  # GB per ms = TB/s
    insts = vb.get("SQ_INSTS_VALU_per_launch", {}).get(k)
    print("| `%s` | %.2f | %.3f | %s | %s | %s | %s | %s | %s |" % (
        k, per_call, ms, "%.2f" % gb if gb else "-", "%.2f" % bw if bw else "-", "%.3f" % (bw / 8.0) if bw else "-",
        "%.2f" % (bw / copy) if bw else "-", vb["VALUBusy_percent"].get(k, "-"), "%.0f" % (insts * per_call / 1e6) if insts else "-"))
print(f"\nSum of kernel times per call: {tot_ms:.3f} ms (under rocprofv3; the side-stream blur overlaps the quadtree, so the wall "
      "time per call is shorter than the sum).")
