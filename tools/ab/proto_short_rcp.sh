#!/bin/bash
# Timing prototype: the reciprocals of the Ant's / Walker's solves (pivots, 2x2 determinants) as rcp + one Newton step
# instead of the IEEE division sequence.  NOT a product form (no range check: wrong bits outside 2^-126 <= |d| <= 2^126).
# Build the variant library here (CPU), then run this script on the GPU box:
#   git apply tools/ab/proto_short_rcp.patch
#   GX_EXTRA_FLAGS_gx_kernels_ant="-DGX_PROTO_SHORT_RCP" GX_EXTRA_FLAGS_gx_kernels_walker="-DGX_PROTO_SHORT_RCP" python tools/build_variant.py shortrcp
#   git apply -R tools/ab/proto_short_rcp.patch
# Result (profiles/r05_ab_short_rcp.log): Ant epoch 302.8 -> 315.3 M (+4.1 %), Walker 160.0 -> 168.5 M (+5.3 %).
# The two bit-safe forms built on it both lost: see profiles/r05_ab_short_rcp.log and DESIGN.md section 10.
cd $GRAFT_REPO_ROOT
V=guardx_amd/lib/variants/libguardx_hip_shortrcp.so
for t in Goal_Ant_8Hazards Goal_Walker_8Hazards; do
  for i in 1 2; do
    python tools/ab_epoch.py $t --reps 5 --tag product
    GX_LIB_EXPERIMENT=1 GX_LIB=$V python tools/ab_epoch.py $t --reps 5 --tag shortrcp
  done
done
