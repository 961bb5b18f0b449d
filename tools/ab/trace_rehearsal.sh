#!/bin/bash
# kernel timeline of the rank-0-of-W rehearsal (tools/rehearse_rank.py) of one robot:
#   tools/ab/trace_rehearsal.sh xmls/ant.xml ant8 [epochs]   -> gpurun_out/trace_<tag>_kernel_trace.csv
robot=$1; tag=$2; epochs=${3:-8}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python3 $GRAFT_REPO_ROOT/tools/rehearse_rank.py --robot $robot --epochs $epochs --warmup 3 > $out/trace_$tag.log 2>&1
f=$(find /tmp/tr_$tag -name "*kernel_trace.csv" | head -1)
cp $f $out/trace_${tag}_kernel_trace.csv
