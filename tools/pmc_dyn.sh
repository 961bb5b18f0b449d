#!/bin/bash
# SQ counter passes over the rollout's dynamics pass (separate --pmc runs, no tracing mixed in):
#   tools/pmc_dyn.sh <tag>   -> gpurun_out/<tag>_dyn_pmc_<set>.csv
tag=${1:-x}
out=$PWD/gpurun_out
export TMPDIR=/tmp
cmd="python3 tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --repeat 6"
run() { # label, counters...
  local label=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmc_${tag}_$label -- $cmd > $out/${tag}_dyn_pmc_$label.log 2>&1
  f=$(find /tmp/pmc_${tag}_$label -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_means.py $f | grep "dyn_tape" | awk -F',' '{print $(NF-8), $(NF-7)}'; else echo "no output for $label"; tail -3 $out/${tag}_dyn_pmc_$label.log; fi
}
run A SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_ANY
run B SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU
run C SQ_IFETCH SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_MISC SQ_INSTS_SENDMSG
