# Same-box A/B of whole trees.  The baselines are rebuilt from git when needed (they are not kept in the tree):
#   mkdir -p _r03 && git archive f93ab85 guardx_amd include bench.py | tar -x -C _r03 && (cd _r03 && python -m guardx_amd.build) && cp tools/ab_epoch.py _r03/tools/
#   (round 4: 3a50373 into _r04; bisection: one directory per commit under _bis/)
set -e
for t in Goal_Swimmer_8Hazards Goal_Point_8Hazards; do
for i in 1 2; do
(cd _r03 && python tools/ab_epoch.py $t --tag r03 2>/dev/null)
GX_LIB_EXPERIMENT=1 GX_LIB=guardx_amd/lib/variants/libguardx_hip_r04.so python tools/ab_epoch.py $t --tag r04 2>/dev/null
python tools/ab_epoch.py $t --tag new 2>/dev/null
done
done
