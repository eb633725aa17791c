cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4
timeout 600 python -m pytest tests/test_projection.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/s4/tests.txt
timeout 900 python bench.py 2>gpurun_out/s4/bench.err | tail -1 > gpurun_out/s4/bench.json
