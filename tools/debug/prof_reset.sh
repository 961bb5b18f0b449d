# sampler kernels: kernel trace + SQ counters of three inline reset() calls (run on the MI355X box through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-reset_prof}
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/profile_reset.py > $O/kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/tools/profile_reset.py > $O/sq.log 2>&1
find $O/kt -name "*kernel_stats.csv" | xargs grep -h "sample_phase\|count_kernel\|scan_kernel\|compact_kernel" | cut -d, -f1-5 | sed 's/(gx::SampleParams.*)"/"/'
find $O/sq -name "*counter_collection.csv" | xargs grep -h "sample_phase2" | awk -F, '{print $(NF-3), $(NF-2)}' | sort | uniq -c | sort -k2 | awk '{print $2, $3}' | sort -u -k1,1
