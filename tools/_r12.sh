mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_gputests_final.log; tail -8 gpurun_out/r04_gputests_final.log
( time python bench.py ) > gpurun_out/r04_v1_bench.json 2> gpurun_out/r04_v1_bench.err; tail -4 gpurun_out/r04_v1_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_v1_bench.json').read().strip().splitlines()[-1])
for k in ("value","value_hbm_resident","value_pcie_inclusive","value_match_popc","config4_frames_per_s","config5_frames_per_s","ms_per_step","exact_checked","single_frame_host_latency_ms","speedup_vs_cpu_1thread"):
    print(k, d.get(k))
print({k:(v['ms'] if isinstance(v,dict) else v) for k,v in d["stages"].items()})
print({k:v for k,v in d["roofline"].items() if k in ("bound","achieved","peak","frac","traffic","launch_ms","kernel")})
print(d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline_all_cores",{}).get("value"), d.get("cpu_baseline_half_of_logical_cpus",{}).get("value"))
print(d["workloads"])
PY
