#!/bin/bash
# LDS bank-conflict share of one kernel (GPU box): rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE on tools/stage_times.py
# usage: tools/pmc_quick.sh k_orient_describe
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/pq
(cd /tmp && B=256 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pq --output-format csv -- python "$OLDPWD/tools/stage_times.py" > /dev/null 2>&1)
python - "$1" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
m = {k: sum(v["v"]) / len(v["v"]) for k, v in acc.items()}
print(sys.argv[1], {k: round(v) for k, v in m.items()}, "conflict share", round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3))
PY
