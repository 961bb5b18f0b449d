#!/bin/bash
# One gx_rollout call (env_num=2000, T=200) under rocprofv3: kernel times, then the PMC passes (separate runs).
#   tools/prof_rollout.sh <tag> [pmc]   -> gpurun_out/<tag>_rollout_N2000_T200_kernel_stats.csv [+ _pmc_*.csv]
set -e
tag=${1:-r04}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
cmd="python3 tools/profile_step.py --mode rollout --env-num 2000 --launches 200"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_roll_$tag -- $cmd > $out/${tag}_prof_rollout.log 2>&1
cp $(find /tmp/prof_roll_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_rollout_N2000_T200_kernel_stats.csv
if [ "$2" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${c}_$tag -- $cmd >> $out/${tag}_prof_rollout.log 2>&1
    python3 tools/pmc_means.py $(find /tmp/pmc_${c}_$tag -name "*counter_collection.csv" | head -1) > $out/${tag}_rollout_N2000_T200_pmc_$c.csv
  done
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d /tmp/pmc_SQ_$tag -- $cmd >> $out/${tag}_prof_rollout.log 2>&1
  python3 tools/pmc_means.py $(find /tmp/pmc_SQ_$tag -name "*counter_collection.csv" | head -1) > $out/${tag}_rollout_N2000_T200_pmc_SQ.csv
fi
