cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s9
B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py > gpurun_out/s9/st.txt 2>&1
B=1024 ORBFE_OVERLAP=0 GEN=S_tum python tools/stage_times.py >> gpurun_out/s9/st.txt 2>&1
for i in 1 2; do timeout 300 python bench.py --no-extras --steps 10 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])' >> gpurun_out/s9/st.txt; done
