#!/bin/bash
# run-to-run spread of the driver-style line's extras on one box, and reset_done_heavy against the round's evidence build
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/noise_bench_$i.json 2>> gpurun_out/noise_err.log
done
python - <<PY
import json
for i in (1,2,3):
    l = json.loads(open(f"gpurun_out/noise_bench_{i}.json").read().strip().splitlines()[-1])
    a = l["api_step_loop_env_steps_per_s"]
    print(i, round(l["value"]/1e6,1), {k: round(v["env_steps_per_s"]/1e6,1) for k,v in l["other_robots"].items() if isinstance(v, dict)}, "rdh", round(l["reset_done_heavy"]["env_steps_per_s"]/1e6,1),
          "api", round(a["value"]/1e6,1), round(a["out_ring_8"]["value"]/1e6,1), "closed", round(l["closed_loop_policy_env_steps_per_s"]/1e6,1), {k: round(v/1e6,1) for k,v in l["closed_loop_policy_wider_env_steps_per_s"].items()},
          "rehearsal", {k: l["multi_gpu_rehearsal"][k]["ms_per_epoch"] for k in ("expand_all","expand_local")}, l["vs_previous_round"]["regressions"])
PY
for i in 1 2; do
python tools/ab_rdh.py new
(cd _bis/5841321 && python tools/ab_rdh.py a39c04bd)
done
