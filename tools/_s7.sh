cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s7
python tools/timeline.py s7/p3 > gpurun_out/s7/p3.txt 2>&1
ORBFE_BENCH_PIPES=2 python tools/timeline.py s7/p2 > gpurun_out/s7/p2.txt 2>&1
