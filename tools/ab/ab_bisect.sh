# Same-box A/B of whole trees.  The baselines are rebuilt from git when needed (they are not kept in the tree):
#   mkdir -p _r03 && git archive f93ab85 guardx_amd include bench.py | tar -x -C _r03 && (cd _r03 && python -m guardx_amd.build) && cp tools/ab_epoch.py _r03/tools/
#   (round 4: 3a50373 into _r04; bisection: one directory per commit under _bis/)
set -e
t=Goal_Swimmer_8Hazards
for i in 1 2; do
(cd _r03 && python tools/ab_epoch.py $t --tag r03 2>/dev/null)
for c in f634cf9 f1597f5 6d4e4cd c975553 29014f5; do
(cd _bis/$c && python tools/ab_epoch.py $t --tag $c 2>/dev/null)
done
python tools/ab_epoch.py $t --tag new 2>/dev/null
done
./tools/probes/rcp_exact_probe
