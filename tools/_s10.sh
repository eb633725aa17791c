cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s10
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/s10/tests.txt
timeout 900 python bench.py 2>gpurun_out/s10/bench.err | tail -1 > gpurun_out/s10/bench.json
python tools/profile_round.py r03_v2 > gpurun_out/s10/prof.txt 2>&1
