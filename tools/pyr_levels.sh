#!/bin/bash
# per-level duration of the chained pyramid launches (GPU box): rocprofv3 kernel trace of tools/stage_times.py, k_pyr_walk
# dispatches grouped by grid size (= level).  usage: [B=1024] tools/pyr_levels.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/pl
(cd /tmp && B=${B:-1024} ORBFE_OVERLAP=0 rocprofv3 --kernel-trace -d /tmp/pl --output-format csv -- python "$OLDPWD/tools/stage_times.py" > /dev/null 2>&1)
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pl/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_pyr_walk" in n or "k_fast_map" in n or "k_blur7" in n or "k_octree" in n or "k_orient" in n:
            key = (n.split("(")[0].replace("void ", "")[:24], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc, key=lambda k: (k[0], -int(k[1] or 0))):
    v = sorted(acc[k])
    print("%-26s grid %-10s wg %-5s calls %3d  median %8.1f us  min %8.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], v[0]))
PY
