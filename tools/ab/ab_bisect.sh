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
