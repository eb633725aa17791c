echo "=== blur + pyramid: separate (0) / in-lane fusion (1) / resize waves beside the blur waves (2)"
for f in 0 2; do
  echo "--- ORBFE_FUSE_BLUR_PYR=$f"
  ORBFE_FUSE_BLUR_PYR=$f B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
done
ORBFE_FUSE_BLUR_PYR=2 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -3
for f in 0 2; do ORBFE_FUSE_BLUR_PYR=$f python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse $f pipes3 value', d['value'])"; done
ORBFE_FUSE_BLUR_PYR=2 B=256 bash tools/pmc_kernel.sh k_blur_pyr "FETCH_SIZE" "WRITE_SIZE" 2>/dev/null
