#!/bin/bash
# usage (on the GPU box): tools/prof.sh <tag>   -> per-kernel averages of one stage_times run
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python $GRAFT_REPO_ROOT/tools/stage_times.py 2>&1 | grep "^[0-9]"
python - <<PY
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/prof_$1/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_"):
        print("%-20s calls %4s avg_us %8.1f min %8.1f max %8.1f" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
