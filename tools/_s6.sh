cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
python tools/timeline.py s6/p3 > gpurun_out/s6/p3.txt 2>&1
ORBFE_BENCH_PIPES=1 python tools/timeline.py s6/p1 > gpurun_out/s6/p1.txt 2>&1
for ov in 0 1 2; do echo "overlap=$ov $(ORBFE_OVERLAP=$ov timeout 300 python bench.py --no-extras --steps 10 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')" >> gpurun_out/s6/res.txt; done
timeout 900 python bench.py 2>gpurun_out/s6/bench.err | tail -1 > gpurun_out/s6/bench.json
