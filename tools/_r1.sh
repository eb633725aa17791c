mkdir -p gpurun_out
( time python bench.py --steps 10 --warmup 2 ) > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -c 600 gpurun_out/r04_bench_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_a.json').read().strip().splitlines()[-1])
for k in ("value","value_hbm_resident","value_pcie_inclusive","value_match_popc","config4_frames_per_s","config5_frames_per_s","ms_per_step"):
    print(k, d.get(k))
print(d["stages"]); print(d["roofline"].get("frac"), d["roofline"].get("launch_ms"))
print(d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_all_cores"))
PY
for fl in "1024 24" "2048 12" "4096 6"; do set -- $fl; echo "== frames $1 launches $2"; python bench.py --steps 6 --warmup 2 --frames $1 --launches $2 --no-extras --seeds 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['stages'])"; done
B=1024 ORBFE_OVERLAP=0 python tools/stage_times.py
B=2048 ORBFE_OVERLAP=0 python tools/stage_times.py
