#!/bin/bash
# kernel timeline of bench.py --gpus 1 over a one-rank RCCL group (GX_FORCE_DIST=1): -> gpurun_out/trace_<tag>_kernel_trace.csv
tag=${1:-rccl1}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
export GX_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 8 --warmup 4 --reps 1 --no-extras --no-cpu-baseline > $out/trace_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
cp $f $out/trace_${tag}_kernel_trace.csv
