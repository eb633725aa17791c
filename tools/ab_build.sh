#!/bin/bash
# A/B of compile-time kernel variants on the GPU box: rebuilds liborbfe.so with each flag set and prints the stage times.
#   usage (GPU box): tools/ab_build.sh "-DQT_MIN_WAVES=6" "-DQT_MIN_WAVES=8" ...      (env B, W, H, NF, GEN as tools/stage_times.py)
# The last build is the default one again (no extra flags), so the tree is left as it was found.
cd "$(dirname "$0")/.."
for flags in "$@" ""; do
    echo "=== flags: '${flags}'"
    if ! ORBFE_EXTRA_FLAGS="$flags" python -c "from orb_slam2_ssd_semantic_amd import _build; _build.build(force=True)"; then
        echo "=== build failed for '${flags}': restoring the default build"
        python -c "from orb_slam2_ssd_semantic_amd import _build; _build.build(force=True)"
        exit 1
    fi
    [ -z "$flags" ] && [ $# -gt 0 ] && break
    python tools/stage_times.py
    [ -n "$AB_CFG5" ] && B=128 W=1920 H=1080 NF=4000 python tools/stage_times.py
done
