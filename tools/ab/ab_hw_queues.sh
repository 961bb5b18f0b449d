#!/bin/bash
# the rank rehearsal of every robot, the one-rank RCCL bench and the 2-rank gloo bench on one box (defaults)
cd $GRAFT_REPO_ROOT
for rb in point swimmer ant walker; do
  echo "== $rb"; python tools/rehearse_rank.py --robot xmls/$rb.xml --epochs 30 2>/dev/null | grep ms_per_epoch | tr -d "\n"; echo
done
for i in 1 2; do
GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2954$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --gpus 1 --steps 10 --warmup 5 --reps 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   rccl1', round(l['value']/1e6,1), {k: round(v['value']/1e6,1) for k,v in l['legs'].items() if isinstance(v, dict)}, round(l['stepping_only']['value']/1e6,1))"
done
