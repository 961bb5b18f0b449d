set -e
for t in Goal_Swimmer_8Hazards Goal_Point_8Hazards; do
for i in 1 2; do
(cd _r03 && python tools/ab_epoch.py $t --tag r03 2>/dev/null)
GX_LIB_EXPERIMENT=1 GX_LIB=guardx_amd/lib/variants/libguardx_hip_r04.so python tools/ab_epoch.py $t --tag r04 2>/dev/null
python tools/ab_epoch.py $t --tag new 2>/dev/null
done
done
