set -e
for t in Goal_Swimmer_8Hazards Goal_Point_8Hazards Goal_Ant_8Hazards; do
for i in 1 2; do
python tools/ab_epoch.py $t --tag cap3072 2>/dev/null
GX_OBS_GRID_CAP=1000000 python tools/ab_epoch.py $t --tag uncapped 2>/dev/null
GX_OBS_GRID_CAP=1536 python tools/ab_epoch.py $t --tag cap1536 2>/dev/null
done
done
