echo "=== FAST with per-lane constants in LDS (150 VGPRs) vs default (157)"
for lib in "" "$PWD/ab/liborbfe_fm_lds.so"; do
  echo "--- ORBFE_LIB=$lib"
  ORBFE_LIB=$lib B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
  for rep in 1 2; do ORBFE_LIB=$lib python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipes3 value', d['value'])"; done
  ORBFE_LIB=$lib ORBFE_BENCH_PIPES=2 python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipes2 value', d['value'])"
  ORBFE_LIB=$lib ORBFE_BENCH_PIPES=4 python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipes4 value', d['value'])"
done
ORBFE_LIB=$PWD/ab/liborbfe_fm_lds.so python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -2
