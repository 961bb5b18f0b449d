# round-2 profile collection (run on the MI355X box through gpurun); summaries are copied into profiles/ by hand
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_prof2
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/roll_kt -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/roll_fetch -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/roll_write -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/roll_sq -- python3 $R/tools/profile_step.py --mode rollout --env-num 2000 --launches 200 > $O/roll_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/step_sq -- python3 $R/tools/profile_step.py --mode step --env-num 2000 --launches 300 > $O/step_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/step_kt -- python3 $R/tools/profile_step.py --mode step --env-num 2000 --launches 300 > $O/step_kt.log 2>&1
find $O -name "*.csv" | wc -l
