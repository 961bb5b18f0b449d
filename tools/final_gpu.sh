#!/bin/bash
# the round's run-time evidence on the final build (through gpurun, from the repo root; the rocprofv3 summaries come
# from tools/collect_profiles.sh):   bash tools/final_gpu.sh r05 [soak|bench]   -> gpurun_out/r05_*  (copy into profiles/)
tag=${1:-r05}; what=${2:-all}
cd $GRAFT_REPO_ROOT
if [ "$what" = "all" ] || [ "$what" = "soak" ]; then
  for rb in point swimmer ant walker; do
    timeout -k 10 500 python tests/soak_parity.py $rb 300000 4096 400 > gpurun_out/${tag}_soak_$rb.log 2>&1 || echo "soak $rb FAILED"
    tail -n 1 gpurun_out/${tag}_soak_$rb.log
    timeout -k 10 400 python tests/soak_variants.py $rb 2 > gpurun_out/${tag}_soak_variants_$rb.log 2>&1 || echo "soak variants $rb FAILED"
    tail -n 1 gpurun_out/${tag}_soak_variants_$rb.log
  done
  # the N > 1 epoch: W ranks in one process, every expanded row of every rank against one engine of W x N envs
  for rb in point swimmer ant walker; do
    timeout -k 10 300 python tests/soak_handoff.py $rb 4 150 256 40 > gpurun_out/${tag}_soak_handoff_$rb.log 2>&1 || echo "soak handoff $rb FAILED"
    tail -n 1 gpurun_out/${tag}_soak_handoff_$rb.log
  done
  for rb in point ant swimmer walker; do
    timeout -k 10 300 python tests/soak_handoff.py $rb 8 60 384 24 > gpurun_out/${tag}_soak_handoff_${rb}_w8.log 2>&1 || echo "soak handoff $rb W=8 FAILED"
    tail -n 1 gpurun_out/${tag}_soak_handoff_${rb}_w8.log
  done
  for rb in ant walker; do   # the robots whose solves changed last (round 5: the pivots' reciprocals): a longer soak
    timeout -k 10 500 python tests/soak_parity.py $rb 1500000 8192 600 > gpurun_out/${tag}_soak_long_$rb.log 2>&1 || echo "long soak $rb FAILED"
    tail -n 1 gpurun_out/${tag}_soak_long_$rb.log
  done
  timeout -k 10 300 python tests/soak_handoff.py point 2 100 2000 200 > gpurun_out/${tag}_soak_handoff_point_w2_full.log 2>&1 || echo "soak handoff point W=2 full size FAILED"
  tail -n 1 gpurun_out/${tag}_soak_handoff_point_w2_full.log
fi
if [ "$what" = "all" ] || [ "$what" = "bench" ]; then
  python -m pytest tests -m gpu -q 2>&1 | tail -n 4 > gpurun_out/${tag}_gputest_final.log
  cat gpurun_out/${tag}_gputest_final.log
  # the RCCL one-rank run leaves its report, RCCL's own log lines and the forced-dist bench line in gpurun_out/
  cp gpurun_out/rccl_one_rank_report.json gpurun_out/${tag}_rccl_one_rank_report.json
  grep "NCCL INFO" gpurun_out/rccl_one_rank.log | grep -v "NET/\|lugin" | head -120 > gpurun_out/${tag}_rccl_one_rank_nccl.log
  cp gpurun_out/bench_force_dist_one_rank.json gpurun_out/${tag}_bench_force_dist_one_rank.json
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
  python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_style.json 2> gpurun_out/bench_err.log
  python bench.py > gpurun_out/${tag}_bench_full.json 2>> gpurun_out/bench_err.log
  GX_BENCH_FORCE_DEVICE=0 GX_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline > gpurun_out/${tag}_rehearsal_2ranks_one_gpu_gloo.json 2>> gpurun_out/bench_err.log
  python - <<PY
import json
for f in ("gpurun_out/${tag}_bench_driver_style.json", "gpurun_out/${tag}_bench_full.json"):
    l = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(l["value"]/1e6,1), l["ms_per_step"], l["roofline"]["frac"], l["roofline"].get("traffic"), l.get("repetitions",{}).get("values"), round(l.get("preconditioned",{}).get("value",0)/1e6,1), l.get("vs_previous_round",{}).get("regressions"),
          {k: round(v["env_steps_per_s"]/1e6,1) for k,v in l["other_robots"].items() if isinstance(v, dict)}, round(l["reset_done_heavy"]["env_steps_per_s"]/1e6,1),
          l["api_step_loop_env_steps_per_s"], l.get("closed_loop_policy_env_steps_per_s"), l.get("closed_loop_policy_wider_env_steps_per_s"),
          {k: l.get("cpu_baseline",{}).get(k) for k in ("value","cores","value_1thread","host")})
    mg = l.get("multi_gpu_rehearsal", {})
    print("  rehearsal:", {k: (mg[k]["ms_per_epoch"], mg[k]["model"]["at_310GBps"]["weak_scaling_efficiency"]) for k in ("expand_all","expand_local") if k in mg}, mg.get("one_gpu_own_sampler"))
l = json.loads(open("gpurun_out/${tag}_rehearsal_2ranks_one_gpu_gloo.json").read().strip().splitlines()[-1])
print("2 ranks / gloo / one GPU:", round(l["value"]/1e6,1), {k: round(v["value"]/1e6,1) for k,v in l["legs"].items() if isinstance(v, dict)}, round(l["stepping_only"]["value"]/1e6,1), round(l["preconditioned"]["value"]/1e6,1))
PY
fi
