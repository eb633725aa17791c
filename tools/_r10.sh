echo "=== FAST run length sweep (blur at 40)"
for r in 40 44 48 52 56 60 68; do
  echo -n "ROWS_FAST=$r  "
  ORBFE_ROWS_FAST=$r B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2 | tr '\n' ' '; echo
done
echo "=== blur run length sweep (FAST at 40)"
for r in 32 36 40 44 48; do
  echo -n "ROWS_BLUR=$r  "
  ORBFE_ROWS_BLUR=$r B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py 2>/dev/null | tail -2 | tr '\n' ' '; echo
done
for r in 40 48 56; do ORBFE_ROWS_FAST=$r python bench.py --steps 8 --warmup 2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows_fast $r pipes3 value', d['value'])"; done
