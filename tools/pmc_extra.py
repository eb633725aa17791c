"""Supplementary counters of the bench command, one rocprofv3 --kernel-trace --pmc pass per counter group (never together
with another trace domain): LDS bank conflicts, LDS wait, memory-unit stall, occupancy, L2 hit rate -- the evidence behind
"LDS-latency bound" (quadtree) and "line-traffic bound" (describe) in DESIGN.md.
usage (GPU box):  python tools/pmc_extra.py <tag> [more bench args]     ->  gpurun_out/<tag>_pmc_extra.json"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1]
bench_args = ["--no-extras", "--launches", "2", "--steps", "4", "--warmup", "2"] + sys.argv[2:]   # e.g. --workload S_tum --seeds 32
cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
env = dict(os.environ, TMPDIR="/tmp")
GROUPS = [["LDSBankConflict"], ["MemUnitStalled"], ["OccupancyPercent"], ["SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"],
          ["SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]]


def kname(n):
    return n.replace("void ", "").split("(")[0].split("<")[0]


res = {}
for grp in GROUPS:
    d = os.path.join(OUT, f"{tag}_pmcx_{grp[0]}")
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp",
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    acc = {c: collections.defaultdict(list) for c in grp}
    if f:
        for row in csv.DictReader(open(f[0])):
            k = kname(row["Kernel_Name"])
            if k.startswith("k_") and row["Counter_Name"] in acc:
                acc[row["Counter_Name"]][k].append(float(row["Counter_Value"]))
    else:
        print("no counter file for", grp, "rc", r.returncode, r.stdout[-400:], file=sys.stderr)
    for c, a in acc.items():
        res[c] = {k: round(sum(v) / len(v), 3) for k, v in a.items()}
    shutil.rmtree(d, ignore_errors=True)
out = {"per_launch_mean": res, "command": " ".join(["python", "bench.py"] + bench_args),
       "note": "one rocprofv3 --kernel-trace --pmc pass per group; derived metrics (LDSBankConflict, MemUnitStalled, OccupancyPercent) in "
               "percent as rocprofv3 defines them; raw SQ / TCC counters summed over the device per launch"}
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res:
    out["L2_hit_rate"] = {k: round(res["TCC_HIT_sum"][k] / max(res["TCC_HIT_sum"][k] + res["TCC_MISS_sum"].get(k, 0.0), 1.0), 4)
                          for k in res["TCC_HIT_sum"]}
if "SQ_WAIT_INST_LDS" in res and "SQ_WAVE_CYCLES" in res:
    out["wave_cycles_waiting_on_LDS_frac"] = {k: round(res["SQ_WAIT_INST_LDS"][k] / max(res["SQ_WAVE_CYCLES"].get(k, 0.0), 1.0), 4)
                                              for k in res["SQ_WAIT_INST_LDS"]}
json.dump(out, open(os.path.join(OUT, f"{tag}_pmc_extra.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in out if k not in ("per_launch_mean", "note", "command")}, indent=1))
print(json.dumps({c: res[c] for c in ("LDSBankConflict", "MemUnitStalled", "OccupancyPercent") if c in res}, indent=1))
