mkdir -p gpurun_out
echo "=== blur lane alignment A/B (ORBFE_BLUR_ALIGN lanes)"
for al in 1 16 32; do
  echo "--- align $al"
  ORBFE_BLUR_ALIGN=$al B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
  ORBFE_BLUR_ALIGN=$al B=256 bash tools/pmc_kernel.sh k_blur7 "FETCH_SIZE" "WRITE_SIZE" 2>/dev/null
done
echo "=== parity with align 16 / 32"
ORBFE_BLUR_ALIGN=16 python -m pytest tests/test_gpu_extract.py -q -x -k "stage_by_stage or parameter_sweep or edge_cases" 2>&1 | tail -2
ORBFE_BLUR_ALIGN=32 python -m pytest tests/test_gpu_extract.py -q -x -k "stage_by_stage or parameter_sweep" 2>&1 | tail -2
echo "=== describe without the LDS patch (DS_GLOBAL_SAMPLES)"
B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
ORBFE_LIB=$PWD/ab/liborbfe_ds_global.so B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
ORBFE_LIB=$PWD/ab/liborbfe_ds_global.so python -m pytest tests/test_gpu_extract.py -q -x -k "stage_by_stage" 2>&1 | tail -2
bash tools/pmc_quick.sh k_orient_describe 2>/dev/null
ORBFE_LIB=$PWD/ab/liborbfe_ds_global.so bash tools/pmc_quick.sh k_orient_describe 2>/dev/null
