#!/bin/bash
# kernel timeline (start/end per kernel, all streams) of a few bench epochs of one task:
#   tools/trace_epochs.sh <task> <tag> [epochs]      -> gpurun_out/trace_<tag>_kernel_trace.csv
task=$1; tag=$2; epochs=${3:-12}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $GRAFT_REPO_ROOT/tools/profile_epochs.py $task $epochs > $out/trace_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
cp $f $out/trace_${tag}_kernel_trace.csv
