#!/bin/bash
# usage (GPU box): tools/ubench/match_variants.sh "<flagset1>" "<flagset2>" ...   e.g. "" "-DBM_SKIP_RANK"
cd $GRAFT_REPO_ROOT
S=orb_slam2_ssd_semantic_amd/csrc
for f in "$@"; do
  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 $f -I include -I $S $S/orbfe_api.hip $S/orbfe_kernels.hip $S/orbfe_match.hip tools/ubench/match_variant_main.cpp -o /tmp/mv 2>/dev/null && echo "variant [$f]" && /tmp/mv | tail -1
done
