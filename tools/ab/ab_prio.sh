set -e
export GX_LIB_EXPERIMENT=1
V=guardx_amd/lib/variants/libguardx_hip_r04.so
for t in Goal_Swimmer_8Hazards Goal_Point_8Hazards Goal_Ant_8Hazards; do
for i in 1 2; do
GX_LIB=$V python tools/ab_epoch.py $t --tag r04 2>/dev/null
python tools/ab_epoch.py $t --tag new 2>/dev/null
done
done
