cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_prof
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_kt -- python3 $R/tools/profile_step.py --mode step --env-num 2000 --launches 300 > $O/step_kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/step_sq -- python3 $R/tools/profile_step.py --mode step --env-num 2000 --launches 300 > $O/step_sq.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/step_ic -- python3 $R/tools/profile_step.py --mode step --env-num 2000 --launches 300 > $O/step_ic.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/roll_sq -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_sq.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O/roll_ic -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_ic.log 2>&1
find $O -name "*.csv" | head -30
