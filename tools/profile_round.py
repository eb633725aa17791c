"""Round profile on the GPU box: rocprofv3 kernel statistics and PMC passes of the bench command, summarised into
gpurun_out/<tag>_* (copy what should be judged into profiles/).

usage (GPU box):  python tools/profile_round.py <tag> [bench args...]
passes (each its own rocprofv3 run, as the microarchitecture guide prescribes: counters never share a run with a trace
domain other than the kernel trace):
   1. --kernel-trace --stats                  -> <tag>_kernel_stats.csv
   2. --kernel-trace --pmc FETCH_SIZE         \
   3. --kernel-trace --pmc WRITE_SIZE         /  -> <tag>_pmc_hbm.json (per kernel, per launch; 2 x FETCH + WRITE corrected bytes)
   4. --kernel-trace --pmc VALUBusy           -> <tag>_pmc_valubusy.json
   5. --kernel-trace --stats of the default command (pipes) -> <tag>_kernel_stats_pipes.csv
"""
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
bench_args = sys.argv[2:] or ["--no-extras", "--launches", "2", "--steps", "4", "--warmup", "2", "--seeds", "32"]   # 32 seeds: made in-process (no worker pool under the profiler)
cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + bench_args
# the per-kernel passes run the step on ONE stream (ORBFE_BENCH_PIPES=1): with the default pipes, kernels of several
# sub-batches share the CUs and a kernel's begin-to-end interval is not its own duration; the default command gets one more
# kernel-trace pass of its own at the end (<tag>_kernel_stats_pipes.csv)
env = dict(os.environ, TMPDIR="/tmp", ORBFE_BENCH_PIPES="1")


def kname(n):
    return n.replace("void ", "").split("(")[0].split("<")[0]


def rocprof(args, sub, env=env, cmd=cmd):
    d = os.path.join(OUT, f"{tag}_{sub}")
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return d, (json.loads(line[-1]) if line else None), r.returncode


d, bench_line, rc = rocprof(["--kernel-trace", "--stats"], "trace")
stats = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(OUT, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        if r["Name"].startswith(("k_", "void k_")):
            print("%-22s calls %5s avg_us %9.1f min %9.1f max %9.1f pct %5s" % (kname(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                         float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
if bench_line:
    json.dump(bench_line, open(os.path.join(OUT, f"{tag}_bench_under_rocprof.json"), "w"), indent=1)

frames = bench_line["config"]["frames_per_launch"] if bench_line else 1024
# the shape the counters belong to: bench.py attaches counter-derived figures to a line only when all five keys match
cfg = bench_line["config"] if bench_line else {}
shape = {"width": cfg.get("width", 640), "height": cfg.get("height", 480), "nfeatures": cfg.get("nfeatures", 1000),
         "workload": cfg.get("workload_name", "S"), "frames_per_launch": frames}


def pmc_multi(counters):
    d, _, _ = rocprof(["--kernel-trace", "--pmc"] + counters, "pmc_" + counters[0])
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    acc = {c: collections.defaultdict(list) for c in counters}
    if f:
        for r in csv.DictReader(open(f[0])):
            k = kname(r["Kernel_Name"])
            if k.startswith("k_") and r["Counter_Name"] in acc:
                acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
    shutil.rmtree(d, ignore_errors=True)
    return {c: {k: {"launches": len(v), "mean_per_launch": round(sum(v) / len(v), 1)} for k, v in a.items()} for c, a in acc.items()}


def pmc(counter):
    return pmc_multi([counter])[counter]


fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
hbm = {"FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "frames_per_launch": frames, "shape": shape,
       "command": " ".join(["python", "bench.py"] + bench_args),
       "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KB per launch as reported. "
               "gfx950 correction calibrated on this repo's access shapes (profiles/r01_fetch_calibration.txt): HBM read "
               "bytes = 2 x FETCH_SIZE, HBM write bytes = WRITE_SIZE.",
       "corrected_hbm_bytes_per_launch": {k: int((2 * fetch[k]["mean_per_launch"] + write.get(k, {"mean_per_launch": 0})["mean_per_launch"]) * 1024)
                                          for k in fetch}}
json.dump(hbm, open(os.path.join(OUT, f"{tag}_pmc_hbm.json"), "w"), indent=1)
vb = pmc("VALUBusy")
sq = pmc_multi(["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"])
json.dump({"VALUBusy_percent": {k: v["mean_per_launch"] for k, v in vb.items()}, "frames_per_launch": frames, "shape": shape,
           "SQ_INSTS_VALU_per_launch": {k: v["mean_per_launch"] for k, v in sq["SQ_INSTS_VALU"].items()},
           "SQ_INSTS_SALU_per_launch": {k: v["mean_per_launch"] for k, v in sq["SQ_INSTS_SALU"].items()},
           "SQ_WAVES_per_launch": {k: v["mean_per_launch"] for k, v in sq["SQ_WAVES"].items()},
           "command": " ".join(["python", "bench.py"] + bench_args),
           "note": "rocprofv3 --kernel-trace --pmc VALUBusy (derived metric) and, in another pass, SQ_INSTS_VALU / SQ_INSTS_SALU / "
                   "SQ_WAVES (wave-level instruction counts); mean over the launches of each kernel"},
          open(os.path.join(OUT, f"{tag}_pmc_valubusy.json"), "w"), indent=1)
print(json.dumps(hbm["corrected_hbm_bytes_per_launch"]))
print(json.dumps({k: v["mean_per_launch"] for k, v in vb.items()}))
print(json.dumps({k: v["mean_per_launch"] for k, v in sq["SQ_INSTS_VALU"].items()}))
shutil.rmtree(os.path.join(OUT, f"{tag}_trace"), ignore_errors=True)

# the default command: several sub-batches in flight (intervals overlap; Calls and TotalDuration are what to read)
env_p = dict(os.environ, TMPDIR="/tmp")
env_p.pop("ORBFE_BENCH_PIPES", None)
cmd_p = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--steps", "4", "--warmup", "2", "--seeds", "32"] if len(sys.argv) <= 2 else cmd
d, line_p, _ = rocprof(["--kernel-trace", "--stats"], "trace_pipes", env=env_p, cmd=cmd_p)
st = glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(OUT, f"{tag}_kernel_stats_pipes.csv"))
if line_p:
    json.dump(line_p, open(os.path.join(OUT, f"{tag}_bench_under_rocprof_pipes.json"), "w"), indent=1)
shutil.rmtree(d, ignore_errors=True)
