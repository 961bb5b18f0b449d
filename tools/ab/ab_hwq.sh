#!/bin/bash
# GPU_MAX_HW_QUEUES (hardware queues per priority class a process's streams share, ROCclr default 4) and the one-rank RCCL bench
cd $GRAFT_REPO_ROOT
export GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
for q in 2 3 4 6; do
GPU_MAX_HW_QUEUES=$q MASTER_PORT=2958$i python bench.py --gpus 1 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', round(l['value']/1e6,1), {k: round(v['value']/1e6,1) for k,v in l['legs'].items() if isinstance(v, dict)}, round(l['stepping_only']['value']/1e6,1))"
done
done
for q in 2 4; do GPU_MAX_HW_QUEUES=$q python tools/rehearse_rank.py --epochs 30 2>/dev/null | grep ms_per_epoch | tr -d "\n"; echo " point W=8 queues $q"; GPU_MAX_HW_QUEUES=$q python tools/ab_epoch.py Goal_Point_8Hazards --reps 5 2>/dev/null | tail -1; done
