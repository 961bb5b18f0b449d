#!/bin/bash
# bench.py --gpus 1 over a one-rank RCCL group: TapeHandoff's probe of the collective's hardware queue on / off, with the
# default number of hardware queues per priority (GPU_MAX_HW_QUEUES, ROCclr: 4) and with 8
cd $GRAFT_REPO_ROOT
export GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
for q in 4 8; do
for p in 1 0; do
GPU_MAX_HW_QUEUES=$q GX_HANDOFF_QUEUE_PROBE=$p MASTER_PORT=2957$i python bench.py --gpus 1 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q probe $p', round(l['value']/1e6,1), {k: round(v['value']/1e6,1) for k,v in l['legs'].items() if isinstance(v, dict)}, round(l['stepping_only']['value']/1e6,1), l['stepping_only'].get('handoff_queue_probe'))"
done
done
done
