#!/bin/bash
# usage (GPU box): tools/pmc.sh "<COUNTERS...>" -> per-kernel mean counter values of one stage_times run
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_tmp
rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_tmp -- python $GRAFT_REPO_ROOT/tools/stage_times.py > /dev/null 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/pmc_tmp/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].replace("void ","").split("(")[0].split("<")[0]
    if k.startswith("k_"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
