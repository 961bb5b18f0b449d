# gx_rollout standalone at env_num=2000, T=200: kernel trace and HBM counters (run on the MI355X box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/roll_prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/sq -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/sq.log 2>&1
grep -h "tape_kernel" $O/kt/*/*kernel_stats.csv | cut -c1-40,150-250
for d in fetch write sq; do grep -h "tape_kernel" $O/$d/*/*counter_collection.csv | awk -F, '{print substr($9,11,16), $(NF-3), $(NF-2)}' | sort | uniq -c | awk '{print $2,$3,$4, "x"$1}' | sort -u; done
