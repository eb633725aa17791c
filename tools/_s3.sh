cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s3
for flags in "-DFM_WAVES_PER_EU=2" ""; do
  ORBFE_EXTRA_FLAGS="$flags" python -c "from orb_slam2_ssd_semantic_amd import _build; _build.build(force=True)" >/dev/null 2>&1 || exit 1
  for P in 1 2 3 4; do
    echo "flags='$flags' P=$P $(ORBFE_BENCH_PIPES=$P timeout 300 python bench.py --no-extras --steps 10 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"])')" >> gpurun_out/s3/res.txt
  done
done
