#!/bin/bash
# the second class of engine streams (hand-off, sampler beside a hand-off) at ordinary priority (default) against least (GX_AUX_PRIORITY=-1), same
# box: the driver-style bench line, the 2-rank gloo rehearsal, the one-rank RCCL bench (GX_FORCE_DIST=1)
cd $GRAFT_REPO_ROOT
LO="GX_AUX_PRIORITY=-1"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_prio_bench_mid.json 2> gpurun_out/ab_prio_err.log
env $LO python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_prio_bench_lo.json 2>> gpurun_out/ab_prio_err.log
GX_BENCH_FORCE_DEVICE=0 GX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline > gpurun_out/ab_prio_2ranks_mid.json 2>> gpurun_out/ab_prio_err.log
env $LO GX_BENCH_FORCE_DEVICE=0 GX_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline > gpurun_out/ab_prio_2ranks_lo.json 2>> gpurun_out/ab_prio_err.log
GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --gpus 1 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline > gpurun_out/ab_prio_rccl1_mid.json 2>> gpurun_out/ab_prio_err.log
env $LO GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29536 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --gpus 1 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline > gpurun_out/ab_prio_rccl1_lo.json 2>> gpurun_out/ab_prio_err.log
python - <<PY
import json
def last(f):
    return json.loads(open(f).read().strip().splitlines()[-1])
for tag in ("mid", "lo"):
    l = last(f"gpurun_out/ab_prio_bench_{tag}.json")
    print(tag, "bench:", round(l["value"]/1e6,1), l["repetitions"]["values"] if "repetitions" in l else None,
          {k: round(v["env_steps_per_s"]/1e6,1) for k,v in l["other_robots"].items() if isinstance(v, dict)}, "rdh", round(l["reset_done_heavy"]["env_steps_per_s"]/1e6,1),
          "api", l["api_step_loop_env_steps_per_s"], "closed", l.get("closed_loop_policy_env_steps_per_s"), l.get("closed_loop_policy_wider_env_steps_per_s"),
          "regr", l.get("vs_previous_round",{}).get("regressions"))
    mg = l.get("multi_gpu_rehearsal", {})
    print("   rehearsal:", {k: mg[k]["ms_per_epoch"] for k in ("expand_all","expand_local") if k in mg}, mg.get("one_gpu_own_sampler"))
    for kind in ("2ranks", "rccl1"):
        l = last(f"gpurun_out/ab_prio_{kind}_{tag}.json")
        print("  ", kind, round(l["value"]/1e6,1), {k: round(v["value"]/1e6,1) for k,v in l["legs"].items() if isinstance(v, dict)}, round(l["stepping_only"]["value"]/1e6,1))
PY
