cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s5
timeout 900 python -m pytest tests/test_projection.py tests/test_shim_ref.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 > gpurun_out/s5/tests.txt
