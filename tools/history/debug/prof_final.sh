# end-of-round evidence (run on the MI355X box through gpurun): tests, bench line, kernel trace of the bench, sampler counters
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; tail -2 $O/gputest.log
python bench.py > $O/bench.json 2> $O/bench.err && cut -c1-160 $O/bench.json
python tools/debug/sampler_waves.py > $O/sampler_waves.log 2>&1
./tools/probes/valu_issue_probe > $O/probe_valu_issue.log 2>&1
./tools/probes/threefry_chain_probe > $O/probe_threefry_chain.log 2>&1
./tools/probes/lds_residency_probe > $O/probe_lds_residency.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_kt -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/bench_kt.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/reset_kt -- python3 $R/tools/profile_reset.py > $O/reset_kt.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/reset_sq -- python3 $R/tools/profile_reset.py > $O/reset_sq.log 2>&1
find $O -name "*.csv" | wc -l
# copy what is judged into profiles/ by hand (gpurun merges into an existing gpurun_out/final: take the newest files)
