echo "=== describe early-out for padding-only workgroups + rows per wave sweep"
for r in 40 56 64 80 112; do
  echo "--- ORBFE_ROWS=$r"
  ORBFE_ROWS=$r B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2
done
for r in 40 64 80; do ORBFE_ROWS=$r python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows $r pipes3 value', d['value'])"; done
python -m pytest tests/test_gpu_extract.py -q -x 2>&1 | tail -2
