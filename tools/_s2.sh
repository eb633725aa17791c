cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s2
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/s2/tests.txt
for P in 1 2 3; do ORBFE_BENCH_PIPES=$P timeout 300 python bench.py --no-extras --steps 10 2>gpurun_out/s2/p$P.err | tail -1 > gpurun_out/s2/p$P.json; done
ORBFE_BENCH_PIPES=2 ORBFE_BENCH_MATCH_STREAM=0 timeout 300 python bench.py --no-extras --steps 10 --no-match 2>/dev/null | tail -1 > gpurun_out/s2/p2_nomatch.json
timeout 300 python bench.py --no-extras --steps 10 --no-match 2>/dev/null | tail -1 > gpurun_out/s2/p1_nomatch.json
