#!/bin/bash
# stream priorities (sampler's side stream: GX_SIDE_PRIORITY, hand-off's stream: GX_AUX_PRIORITY; -1 least, 0 middle, 1 highest):
# the rank rehearsal at W = 8 and the one-GPU epochs
cd $GRAFT_REPO_ROOT
run() { rb=$1; tag=$2; shift; shift; env "$@" python tools/rehearse_rank.py --robot xmls/$rb.xml --epochs 30 > gpurun_out/reh_${rb}_$tag.json; echo "== $rb $tag"; grep "ms_per_epoch" gpurun_out/reh_${rb}_$tag.json | tr -d '\n'; echo; }
for rb in ant point swimmer; do
  run $rb lo_lo GX_NOP=1
  run $rb mid_lo GX_SIDE_PRIORITY=0
  run $rb mid_mid GX_SIDE_PRIORITY=0 GX_AUX_PRIORITY=0
  run $rb lo_mid GX_AUX_PRIORITY=0
  run $rb hi_mid GX_SIDE_PRIORITY=1 GX_AUX_PRIORITY=0
done
for t in Goal_Point_8Hazards Goal_Swimmer_8Hazards Goal_Ant_8Hazards; do
  python tools/ab_epoch.py $t --reps 5 --tag side_lo
  GX_SIDE_PRIORITY=0 python tools/ab_epoch.py $t --reps 5 --tag side_mid
  python tools/ab_epoch.py $t --reps 5 --tag side_lo
  GX_SIDE_PRIORITY=0 python tools/ab_epoch.py $t --reps 5 --tag side_mid
done
