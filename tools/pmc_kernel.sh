#!/bin/bash
# Counters of one kernel on the batched extractor (GPU box): one rocprofv3 --pmc pass per counter group over tools/stage_times.py
# (B frames, default 256).  usage: tools/pmc_kernel.sh k_orient_describe "GROUP1 COUNTERS" "GROUP2 COUNTERS" ...
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
K="$1"; shift
for grp in "$@"; do
  rm -rf /tmp/pk
  (cd /tmp && B=${B:-256} ORBFE_OVERLAP=0 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pk --output-format csv -- python "$OLDPWD/tools/stage_times.py" > /dev/null 2>&1)
  python - "$K" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pk/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v) / len(v), 2) for k, v in acc.items()})
PY
done
