cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_prof_legs
rm -rf $O; mkdir -p $O
for rb in ant walker; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${rb}_kt -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --robot xmls/$rb.xml > $O/${rb}_kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/${rb}_sq -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --robot xmls/$rb.xml > $O/${rb}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES --output-format csv -d $O/${rb}_mem -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 --robot xmls/$rb.xml > $O/${rb}_mem.log 2>&1
done
find $O -name "*.csv" | wc -l
