#!/bin/bash
# usage (GPU box): tools/ubench/match_variants.sh ["<extra hipcc flags>" ...]
# Builds the product sources together with match_variant_main.cpp (once per flag set, e.g. "" "-DSOME_EXPERIMENT") and
# prints the sustained time per orbfe_match_bf_frames_device call (256 pairs of 1004 x 1004, 500 calls).
cd $GRAFT_REPO_ROOT
S=orb_slam2_ssd_semantic_amd/csrc
[ $# -eq 0 ] && set -- ""
for f in "$@"; do
  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 $f -I include -I $S $S/orbfe_api.hip $S/orbfe_kernels.hip $S/orbfe_match.hip tools/ubench/match_variant_main.cpp -o /tmp/mv 2>/dev/null && echo "variant [$f]" && /tmp/mv | tail -1
done
