cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s8
timeout 900 python -m pytest tests/test_gpu_extract.py -q -m gpu -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -5 > gpurun_out/s8/tests.txt
ORBFE_OVERLAP=0 bash tools/ab_build.sh "-DBL_MIN_WAVES=7" "-DBL_MIN_WAVES=8" > gpurun_out/s8/ab.txt 2>&1
ORBFE_OVERLAP=0 python tools/stage_times.py >> gpurun_out/s8/ab.txt 2>&1
