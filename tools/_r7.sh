echo "=== fused blur + pyramid pass (ORBFE_FUSE_BLUR_PYR=1) vs separate"
for f in 0 1; do
  echo "--- fuse $f"
  ORBFE_FUSE_BLUR_PYR=$f B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
  ORBFE_FUSE_BLUR_PYR=$f B=1024 python tools/stage_times.py 2>/dev/null | tail -1
done
ORBFE_FUSE_BLUR_PYR=1 python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -3
for f in 0 1; do ORBFE_FUSE_BLUR_PYR=$f python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse $f pipes3 value', d['value'])"; done
