set -e
export GX_LIB_EXPERIMENT=1
V=guardx_amd/lib/variants/libguardx_hip_swim_noilp.so
for i in 1 2 3; do
python tools/ab_epoch.py Goal_Swimmer_8Hazards --tag ilp
GX_LIB=$V python tools/ab_epoch.py Goal_Swimmer_8Hazards --tag noilp
done
python tools/ab_epoch.py Goal_Swimmer_8Hazards --tag ilp --alone
GX_LIB=$V python tools/ab_epoch.py Goal_Swimmer_8Hazards --tag noilp --alone
