mkdir -p gpurun_out
python tools/profile_round.py r04_v1 2>&1 | tail -25
