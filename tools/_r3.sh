echo "=== blur 64-byte pieces A/B"
for pc in 0 1; do
  echo "--- ORBFE_BLUR_PIECES=$pc"
  ORBFE_BLUR_PIECES=$pc B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
  ORBFE_BLUR_PIECES=$pc B=256 bash tools/pmc_kernel.sh k_blur7 "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES" 2>/dev/null
done
python -m pytest tests/test_gpu_extract.py tests/test_gpu_blur_rounding.py -q -x 2>&1 | tail -2
