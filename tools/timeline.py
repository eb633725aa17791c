"""GPU timeline of the bench command (GPU box): rocprofv3 --kernel-trace, then from the dispatch intervals
   - the share of the wall interval in which at least one kernel was running (idle gaps),
   - the mean number of kernels in flight,
   - per kernel: calls, sum of durations, share of the busy time.
usage: python tools/timeline.py <tag> [bench args...]   -> gpurun_out/<tag>_timeline.json"""
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
args = sys.argv[2:] or ["--no-extras", "--steps", "6", "--warmup", "2"]
d = os.path.join(OUT, tag + "_tl")
shutil.rmtree(d, ignore_errors=True)
r = subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py")] + args,
                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
bench = json.loads(line[-1]) if line else None
f = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))[0]
ev = []
for row in csv.DictReader(open(f)):
    ev.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"].replace("void ", "").split("(")[0].split("<")[0]))
ev.sort()
# the timed region = the last steps*launches k_fast_map dispatches and everything between them
fm = [e for e in ev if e[2] == "k_fast_map"]
nl = bench["config"]["frames_per_gpu_per_step"] // bench["config"]["frames_per_launch"] if bench else 24
n_timed = (bench["steps"] if bench else 6) * nl
n_warm = (bench["warmup"] if bench else 2) * nl
t0 = fm[n_warm][0]                                   # the first dispatches are the warm-up steps
t_next = fm[n_warm + n_timed][0] if len(fm) > n_warm + n_timed else None   # what bench.py launches after the timed region
sel = [e for e in ev if e[0] >= t0 and (t_next is None or e[0] < t_next)]
t1 = max(e[1] for e in sel)
pts = sorted([(s, 1) for s, e, _ in sel] + [(e, -1) for s, e, _ in sel])
busy = 0
conc_time = {}
depth, last = 0, t0
for t, dlt in pts:
    if depth > 0:
        busy += t - last
    conc_time[depth] = conc_time.get(depth, 0) + (t - last)
    depth += dlt
    last = t
per = {}
for s, e, k in sel:
    p = per.setdefault(k, [0, 0])
    p[0] += 1
    p[1] += e - s
wall = t1 - t0
res = {"wall_ms": wall / 1e6, "busy_frac": busy / wall, "mean_kernels_in_flight": sum(v[1] for v in per.values()) / wall,
       "time_share_by_kernels_in_flight": {str(k): round(v / wall, 4) for k, v in sorted(conc_time.items())},
       "per_kernel": {k: {"calls": v[0], "sum_ms": round(v[1] / 1e6, 3), "avg_us": round(v[1] / v[0] / 1e3, 1)} for k, v in sorted(per.items(), key=lambda x: -x[1][1])},
       "bench_value": bench["value"] if bench else None, "command": "python bench.py " + " ".join(args),
       "env": {k: os.environ[k] for k in os.environ if k.startswith("ORBFE_")}}
json.dump(res, open(os.path.join(OUT, tag + "_timeline.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
shutil.rmtree(d, ignore_errors=True)
