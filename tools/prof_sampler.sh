#!/bin/bash
# Sampler / rollout kernel times on the GPU box (run through gpurun from the repo root):
#   tools/prof_sampler.sh <tag>   -> gpurun_out/<tag>_sampler_kernel_stats.csv, <tag>_bench_noextras.json, ...
set -e
tag=${1:-r04}
out=$PWD/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_reset_$tag -- python3 tools/profile_reset.py > $out/${tag}_prof_reset.log 2>&1
cp $(find /tmp/prof_reset_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_sampler_kernel_stats.csv
python3 bench.py --no-cpu-baseline --no-extras > $out/${tag}_bench_noextras.json 2> $out/${tag}_bench_noextras.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench_$tag -- python3 bench.py --no-cpu-baseline --no-extras > $out/${tag}_prof_bench.log 2>&1
cp $(find /tmp/prof_bench_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_noextras_kernel_stats.csv
