#!/bin/bash
for r in 1 2; do
for v in old new; do
  cp ab/$v.so orb_slam2_ssd_semantic_amd/liborbfe.so
  echo "== $v"
  B=1024 python tools/stage_times.py 2>&1 | grep -v amdgpu.ids
done
done
