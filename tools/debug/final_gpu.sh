set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1
echo collected
for rb in point swimmer ant walker; do
  timeout -k 10 500 python tests/soak_parity.py $rb 300000 4096 400 > gpurun_out/r03_soak_$rb.log 2>&1
  tail -1 gpurun_out/r03_soak_$rb.log
done
