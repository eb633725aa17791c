echo "=== describe: keypoints per workgroup 16 / 32 / 48"
for v in "" kpw32 kpw48; do
  lib=""; [ -n "$v" ] && lib="$PWD/ab/liborbfe_$v.so"
  echo "--- $v"
  ORBFE_LIB=$lib B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
  ORBFE_LIB=$lib python bench.py --steps 8 --warmup 2 --no-extras --seeds 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipes3 value', d['value'])"
done
ORBFE_LIB=$PWD/ab/liborbfe_kpw32.so python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -2
