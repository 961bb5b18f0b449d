# everything the round's evidence consists of, on the final build (through gpurun from the repo root):
#   bash tools/debug/final_gpu.sh   -> gpurun_out/r03_*  (copy into profiles/)
set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03 > gpurun_out/r03_collect.log 2>&1
echo collected
for rb in point swimmer ant walker; do
  timeout -k 10 500 python tests/soak_parity.py $rb 300000 4096 400 > gpurun_out/r03_soak_$rb.log 2>&1
  tail -1 gpurun_out/r03_soak_$rb.log
done
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r03_gputest_final.log
cat gpurun_out/r03_gputest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_driver_style.json 2> gpurun_out/bench_err.log
python bench.py > gpurun_out/r03_bench_full.json 2>> gpurun_out/bench_err.log
python - <<'PY'
import json
for f in ("gpurun_out/r03_bench_driver_style.json", "gpurun_out/r03_bench_full.json"):
    l = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(l["value"]/1e6,1), l["ms_per_step"], l["roofline"]["frac"], l["roofline"].get("traffic"), round(l.get("cold_start",{}).get("value",0)/1e6,1),
          {k: round(v["env_steps_per_s"]/1e6,1) for k,v in l["other_robots"].items()}, round(l["reset_done_heavy"]["env_steps_per_s"]/1e6,1),
          round(l["api_step_loop_env_steps_per_s"]/1e6,1), l.get("cpu_baseline",{}).get("value"))
PY
