#!/bin/bash
# A/B of two builds of liborbfe.so on one box: bash ab/run.sh [pytest args]
cp ab/new.so orb_slam2_ssd_semantic_amd/liborbfe.so
python -m pytest tests/test_gpu_extract.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2; do
for v in old new; do
  cp ab/$v.so orb_slam2_ssd_semantic_amd/liborbfe.so
  echo "== $v"
  B=1024 python tools/stage_times.py 2>&1 | grep -v amdgpu.ids
done
done
