#!/bin/bash
# same-box A/B of the key-staging ring: four slots + an event wait per rollout (old: a copy of the product library taken
# before the change, guardx_amd/lib/variants/libguardx_hip_old.so) against the long ring without events (new)
cd $GRAFT_REPO_ROOT
OLD=guardx_amd/lib/variants/libguardx_hip_old.so
for t in Goal_Point_8Hazards Goal_Swimmer_8Hazards Goal_Ant_8Hazards Goal_Walker_8Hazards; do
  python tools/ab_epoch.py $t --reps 5 --tag new
  GX_LIB_EXPERIMENT=1 GX_LIB=$OLD python tools/ab_epoch.py $t --reps 5 --tag old
  python tools/ab_epoch.py $t --reps 5 --tag new
  GX_LIB_EXPERIMENT=1 GX_LIB=$OLD python tools/ab_epoch.py $t --reps 5 --tag old
done
for rb in ant point; do
  python tools/rehearse_rank.py --robot xmls/$rb.xml --epochs 30 > gpurun_out/reh_${rb}_new.json
  GX_LIB_EXPERIMENT=1 GX_LIB=$OLD python tools/rehearse_rank.py --robot xmls/$rb.xml --epochs 30 > gpurun_out/reh_${rb}_old.json
done
grep -H "ms_per_epoch\|\"step\"" gpurun_out/reh_ant_new.json gpurun_out/reh_ant_old.json gpurun_out/reh_point_new.json gpurun_out/reh_point_old.json
